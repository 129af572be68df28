"""Per-launch times of HieCoAtten's three per-sample products (B = 256) on the per-sample-tile kernel and on the round-4 path
(library option gemm_f32_sample = 0), interleaved in one process, hipEvent-timed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vqa_amd
ops = vqa_amd.ops
vqa_amd.lib.load()
NS, L, D, E = 256, 196, 2048, 512
g = torch.Generator().manual_seed(1)
M = NS * L
imgf = torch.relu(torch.randn((M, D), generator=g)).cuda()
img = torch.relu(torch.randn((M, E), generator=g)).cuda()
w_emb, b_emb = (torch.randn((E, D), generator=g) * 0.03).cuda(), torch.randn(E, generator=g).cuda()
Wi, bi = (torch.randn((2 * E, E), generator=g) * 0.05).cuda(), torch.randn(2 * E, generator=g).cuda()
dCI = torch.randn((M, 2 * E), generator=g).cuda()
cases = (("img_emb fwd 50176x512x2048", lambda: ops.gemm_rows(imgf, w_emb, L, bias=b_emb, relu=True), 2.0 * M * E * D),
         ("[Cv|img_] fwd 50176x1024x512", lambda: ops.gemm_rows(img, Wi, L, bias=bi), 2.0 * M * 2 * E * E),
         ("dimg dgrad 50176x512x1024", lambda: ops.gemm_rows(dCI, Wi, L, tb=True), 2.0 * M * 2 * E * E))
for name, fn, fl in cases:
    ms = {0: [], 1: []}
    for rnd in range(3):
        for opt in (0, 1):
            with ops.options(gemm_f32_sample=opt):
                fn()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(10):
                    fn()
                b.record()
                torch.cuda.synchronize()
                ms[opt].append(a.elapsed_time(b) / 10)
    for opt in (0, 1):
        best = min(ms[opt])
        print("%-30s %s: %.3f ms = %.1f TF (%.3f of 157.3)" % (name, "per-sample tiles" if opt else "round-4 path    ", best, fl / best / 1e9, fl / best / 1e9 / 157.3))
