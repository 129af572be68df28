"""gemm_rows (per-sample tiles) against gemm on the exact operands of HieCoAtten's three per-sample products at a given batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vqa_amd
ops = vqa_amd.ops
vqa_amd.lib.load()
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
L, D, E = 196, 2048, 512
g = torch.Generator().manual_seed(1)
M = NS * L
imgf = torch.relu(torch.randn((M, D), generator=g)).cuda()
w_emb, b_emb = (torch.randn((E, D), generator=g) * 0.03).cuda(), torch.randn(E, generator=g).cuda()
Wi, bi = (torch.randn((2 * E, E), generator=g) * 0.05).cuda(), torch.randn(2 * E, generator=g).cuda()
dCI = torch.randn((M, 2 * E), generator=g).cuda()
for name, fn in (("img_emb fwd", lambda f: f(imgf, w_emb, bias=b_emb, relu=True)),
                 ("CI fwd", lambda f: f(torch.relu(imgf[:, :E].contiguous()), Wi, bias=bi)),
                 ("dimg dgrad", lambda f: f(dCI, Wi, tb=True))):
    a = fn(lambda *x, **k: ops.gemm_rows(x[0], x[1], L, **k))
    b = fn(lambda *x, **k: ops.gemm(x[0], x[1], splitk=False, **k))
    d = (a != b)
    rows = sorted(set((d.any(1).nonzero().flatten() % L).tolist()))
    print("%-12s NS=%d: %d differing elements; rows (mod L) %s; cols %s; max |diff| %.3e" % (
        name, NS, int(d.sum()), rows[:10], sorted(set(d.any(0).nonzero().flatten().tolist()))[:10], float((a - b).abs().max())))
