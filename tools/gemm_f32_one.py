"""The image-projection GEMM (fp32), a few launches (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vqa_amd
ops = vqa_amd.ops
vqa_amd.lib.load()
M, N, K = 100352, 5000, 2048
g = torch.Generator().manual_seed(0)
A = torch.relu(torch.randn((M, K), generator=g)).cuda()
B = (torch.randn((N, K), generator=g) * 0.03).cuda()
bias = torch.zeros(N, device="cuda")
out = torch.empty((M, N), device="cuda")
for _ in range(3):
    ops.gemm(A, B, bias=bias, out=out)
torch.cuda.synchronize()
