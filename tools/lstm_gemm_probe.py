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
