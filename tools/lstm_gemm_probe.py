import sys; sys.path.insert(0, "/root/repo")
import torch, vqa_amd
ops = vqa_amd.ops
def timed(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
h = torch.randn(512, 1024, device="cuda"); w = torch.randn(4096, 1024, device="cuda"); out = torch.zeros(512, 4096, device="cuda")
dg = torch.randn(512, 4096, device="cuda")
for sk in (True, False):
    print("fwd  splitk", sk, timed(lambda: ops.gemm(h, w, out=out, accumulate=True, splitk=sk)), "us")
    print("dgrad splitk", sk, timed(lambda: ops.gemm(dg, w, tb=True, splitk=sk)), "us")

print("--- other small-M shapes (default routing)")
for (ta, tb, M, N, K) in [(0, 0, 512, 5000, 2048), (0, 0, 512, 5000, 4096), (0, 0, 7168, 1024, 1024), (0, 0, 512, 1000, 1024),
                          (0, 1, 512, 2048, 5000), (0, 1, 7168, 1024, 1024)]:
    A = torch.randn((K, M) if ta else (M, K), device="cuda"); Bm = torch.randn((K, N) if tb else (N, K), device="cuda")
    us = timed(lambda: ops.gemm(A, Bm, ta=bool(ta), tb=bool(tb)))
    print((ta, tb, M, N, K), "%.1f us  %.1f TF" % (us, 2.0 * M * N * K / us / 1e6))
