import sys, os
sys.path.insert(0, "/root/repo")
import torch, vqa_amd
ops = vqa_amd.ops
M, N, K = 100352, 5000, 2048
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); out = torch.empty(M, N, device="cuda")
def timed(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
xs = torch.as_strided(x, (M, K), (4, 1)); ws_ = torch.as_strided(w, (N, K), (4, 1))
for name, a_, b_ in (("real", x, w), ("X aliased (lda=4)", xs, w), ("X and W aliased", xs, ws_), ("W aliased", x, ws_)):
    ms = timed(lambda: ops.gemm(a_, b_, out=out))
    print("%-22s %.3f ms  %.1f TF" % (name, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
