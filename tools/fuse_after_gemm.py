"""Why the fusion kernels take longer inside the step than alone (bench line: 0.53 / 1.02 ms in-step, 0.48 / 0.87 ms alone).
Times mfb_fuse_fwd / mfb_fuse_bwd at the headline shape (a) back to back, (b) each right behind the 14.6-ms image-projection
GEMM, as in the step, (c) behind an HBM-bound launch of the same length instead; and two yardsticks on the same buffers: a plain
device copy of P (2 GB read + 2 GB written, the backward's mix) and a read-only pass (torch.sum, the forward's mix)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vqa_amd
ops = vqa_amd.ops
N, L, O = 512, 196, 1000
g = torch.Generator(device="cuda").manual_seed(7)
X = torch.randn((N * L, 2048), device="cuda", generator=g).relu_()
W = torch.randn((5 * O, 2048), device="cuda", generator=g) * 0.02
P = ops.gemm(X, W, False, False)
q = torch.randn((N, 5 * O), device="cuda", generator=g); pb = torch.randn(5 * O, device="cuda", generator=g)
dY = torch.randn((N * L, O), device="cuda", generator=g)
P2 = torch.empty_like(P)


def ev():
    return torch.cuda.Event(enable_timing=True)


def timed(fn, kernel, before=None, reps=8, sync=True):
    """mean time of `fn`: of the library kernel `kernel` inside it (hipEvent brackets of the library: the op also runs small
    reducers) or, kernel None, of the whole call (torch events)."""
    tot = 0.0
    ops.prof_reset()
    for _ in range(reps):
        if before is not None:
            before()
        ops.prof_enable(kernel is not None)
        a, b = ev(), ev()
        a.record(); fn(); b.record()
        ops.prof_enable(False)
        if sync:
            torch.cuda.synchronize()
            tot += a.elapsed_time(b)
    torch.cuda.synchronize()
    if kernel is not None:
        n, ms = ops.prof_report()[kernel][:2]
        return ms / n
    return tot / reps


fwd = lambda: ops.mfb_fuse_fwd(P, q, N, L, O, seed=123, p_drop=0.1, pbias=pb, normalise=False)
Y, norm, inv, _ = fwd()
dl = torch.randn((N * L, 2), device="cuda", generator=g); ln = torch.randn((N * L, 2), device="cuda", generator=g)
bwd = lambda: ops.mfb_fuse_bwd(dY, Y, norm, inv, P, q, N, L, O, seed=123, p_drop=0.1, want_dbias=True, pbias=pb, lin=(dl, ln))   # the step's folded form
gemm = lambda: ops.gemm(X, W, False, False)
copy7 = lambda: [P2.copy_(P) for _ in range(17)]          # ~14 ms of HBM-bound work
for name, fn, kernel, nbytes in (("mfb_fuse_fwd", fwd, "mfb_fuse_fwd", 2441617408.0), ("mfb_fuse_bwd", bwd, "mfb_fuse_bwd", 4872192000.0),
                                 ("copy P -> P2 (2 GB + 2 GB)", lambda: P2.copy_(P), None, 2.0 * P.numel() * 4),
                                 ("sum(P) (2 GB read)", lambda: P.sum(), None, 1.0 * P.numel() * 4)):
    for _ in range(3):
        fn()
    if kernel is not None:
        t_b2b = timed(fn, kernel, sync=False)
    else:
        a, b = ev(), ev()
        a.record()
        for _ in range(8):
            fn()
        b.record(); torch.cuda.synchronize()
        t_b2b = a.elapsed_time(b) / 8
    t_alone = timed(fn, kernel)
    t_gemm = timed(fn, kernel, before=gemm)
    t_copy = timed(fn, kernel, before=copy7)
    print("%-28s 8 back to back %.4f ms (%.2f TB/s) | chip idle before %.4f ms (%.2f TB/s) | behind the image-projection GEMM "
          "%.4f ms (%.2f TB/s) | behind 14 ms of copies %.4f ms (%.2f TB/s)"
          % (name, t_b2b, nbytes / t_b2b / 1e9, t_alone, nbytes / t_alone / 1e9, t_gemm, nbytes / t_gemm / 1e9,
             t_copy, nbytes / t_copy / 1e9), flush=True)
