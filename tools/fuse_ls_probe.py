"""Rows-per-block probe of the fusion kernels (library options fuse_ls / fuse_ls_bwd = row subsets per sample)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vqa_amd
ops = vqa_amd.ops
N, L, O = 512, 196, 1000
P = torch.randn(N * L, 5 * O, device="cuda"); q = torch.randn(N, 5 * O, device="cuda"); pb = torch.randn(5 * O, device="cuda")
for ls in ("1", "2", "3", "4", "6", "8", "14"):
    ops.set_option("fuse_ls", int(ls))
    ops.set_option("fuse_ls_bwd", int(ls))
    ops.prof_reset(); ops.prof_enable(True)
    for _ in range(6):
        Y, norm, inv, _ = ops.mfb_fuse_fwd(P, q, N, L, O, seed=123, p_drop=0.1, pbias=pb)
        dY = torch.ones_like(Y)
        ops.mfb_fuse_bwd(dY, Y, norm, inv, P, q, N, L, O, seed=123, p_drop=0.1, want_dbias=True, pbias=pb)
    torch.cuda.synchronize()
    rep = ops.prof_report(); ops.prof_enable(False)
    print("LS=%-3s fwd %.4f ms  bwd %.4f ms" % (ls, rep["mfb_fuse_fwd"][1] / rep["mfb_fuse_fwd"][0], rep["mfb_fuse_bwd"][1] / rep["mfb_fuse_bwd"][0]), flush=True)
