"""Timing of the MFB fusion kernels alone at the headline shape (N=512, L=196, O=1000): direct (strided) vs coalesced
(LDS-transposed) P / dP access (library option fuse_coal), fp32 and bf16 projection storage, with and without Philox dropout."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vqa_amd
ops = vqa_amd.ops
N, L, O = 512, 196, 1000
P = torch.randn(N * L, 5 * O, device="cuda"); q = torch.randn(N, 5 * O, device="cuda"); pb = torch.randn(5 * O, device="cuda")
Pb = P.to(torch.bfloat16)
dY = torch.randn((N * L, O), device="cuda", generator=torch.Generator(device="cuda").manual_seed(7))   # ONE seeded dY for every variant
res = {}
for mode, Pm, kw in (("fp32 P", P, {}), ("bf16 P/dP", Pb, {"dp_bf16": True})):
    for pd in (0.1, 0.0):
        for coal in ("0", "1", "2"):
            ops.set_option("fuse_coal", int(coal))
            ops.prof_reset(); ops.prof_enable(True)
            for _ in range(6):
                Y, norm, inv, _ = ops.mfb_fuse_fwd(Pm, q, N, L, O, seed=123, p_drop=pd, pbias=pb)
                out = ops.mfb_fuse_bwd(dY, Y, norm, inv, Pm, q, N, L, O, seed=123, p_drop=pd, want_dbias=True, pbias=pb, **kw)
            torch.cuda.synchronize()
            rep = ops.prof_report()
            ops.prof_enable(False)
            res[(mode, pd, coal)] = (Y.clone(), out[0].clone(), out[1].clone())
            print("%-10s p_drop=%.1f coalesced=%s  fwd %.4f ms  bwd %.4f ms" % (
                mode, pd, coal, rep["mfb_fuse_fwd"][1] / rep["mfb_fuse_fwd"][0], rep["mfb_fuse_bwd"][1] / rep["mfb_fuse_bwd"][0]), flush=True)
        a, b = res[(mode, pd, "0")], res[(mode, pd, "2")]
        print("   identical results (0 vs 2): Y %s dP %s dq %s" % (torch.equal(a[0], b[0]), torch.equal(a[1].view(torch.uint8), b[1].view(torch.uint8)), torch.equal(a[2], b[2])))
