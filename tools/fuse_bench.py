"""Timing of the MFB fusion kernels alone at the headline shape (N=512, L=196, O=1000)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vqa_amd
ops = vqa_amd.ops
N, L, O = 512, 196, 1000
P = torch.randn(N * L, 5 * O, device="cuda"); q = torch.randn(N, 5 * O, device="cuda"); pb = torch.randn(5 * O, device="cuda")
def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for pd in (0.1, 0.0):
    ops.prof_reset(); ops.prof_enable(True)
    for _ in range(5):
        Y, norm, inv, _ = ops.mfb_fuse_fwd(P, q, N, L, O, seed=123, p_drop=pd, pbias=pb)
        dY = torch.randn_like(Y)
        ops.mfb_fuse_bwd(dY, Y, norm, inv, P, q, N, L, O, seed=123, p_drop=pd, want_dbias=True, pbias=pb)
    torch.cuda.synchronize()
    rep = ops.prof_report()
    ops.prof_enable(False)
    for k in ("mfb_fuse_fwd", "mfb_fuse_bwd", "scale_rows", "rowdot"):
        n, ms = rep[k]; print("p_drop=%.1f %-14s %.4f ms" % (pd, k, ms / n))
