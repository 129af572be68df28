"""One bf16 GEMM shape, a few launches (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vqa_amd
ops = vqa_amd.ops
vqa_amd.lib.load()
M, N, K = 100352, 5000, 2048
A = (torch.rand((M, K), device="cuda") * 2 - 1).to(torch.bfloat16)
B = (torch.rand((N, K), device="cuda") * 2 - 1).to(torch.bfloat16)
out = torch.empty((M, N), device="cuda")
for _ in range(3):
    ops.gemm_bf16(A, B, out=out)
torch.cuda.synchronize()
