import sys; sys.path.insert(0, "/root/repo")
import torch, vqa_amd
ops = vqa_amd.ops
M, N, K = 100352, 5000, 2048
A = (torch.rand((M, K), device="cuda") * 2 - 1).to(torch.bfloat16); B = (torch.rand((N, K), device="cuda") * 2 - 1).to(torch.bfloat16)
bias = torch.zeros(N, device="cuda")
def timed(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
of = torch.empty((M, N), device="cuda"); ob = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
for _ in range(2):
    print("fp32 out %.3f ms" % timed(lambda: ops.gemm_bf16(A, B, bias=bias, out=of)))
    print("bf16 out %.3f ms" % timed(lambda: ops.gemm_bf16(A, B, bias=bias, out=ob, out_bf16=True)))
