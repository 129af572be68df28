"""bf16 GEMM timing on the hot shapes (random operands).  VQF_GEMM_BF16_BIG=0 selects the 128x128 kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vqa_amd
ops = vqa_amd.ops
vqa_amd.lib.load()


def timed(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


shapes = [("img fwd   (0,0)", 0, 0, 100352, 5000, 2048), ("coatt fwd (0,0)", 0, 0, 100352, 1024, 1024),
          ("img wgrad (1,1)", 1, 1, 5000, 2048, 100352), ("coatt wgrad(1,1)", 1, 1, 1024, 1024, 100352),
          ("coatt dgrad(0,1)", 0, 1, 100352, 1024, 1024)]
for name, ta, tb, M, N, K in shapes:
    A = (torch.rand((K, M) if ta else (M, K), device="cuda") * 2 - 1).to(torch.bfloat16)
    B = (torch.rand((K, N) if tb else (N, K), device="cuda") * 2 - 1).to(torch.bfloat16)
    out = torch.empty((M, N), device="cuda")
    ms = timed(lambda: ops.gemm_bf16(A, B, ta=bool(ta), tb=bool(tb), out=out))
    print("%-18s M=%6d N=%5d K=%6d  %.3f ms  %.0f TFLOP/s" % (name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
    del A, B, out
