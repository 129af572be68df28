import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vqa_amd
ops = vqa_amd.ops; vqa_amd.lib.load()
M = 512 * 196
g = torch.Generator().manual_seed(0)
Yp = (torch.rand(M, 1024, generator=g) * 2 - 1).cuda()
Y = Yp[:, :1000].contiguous()
W = (torch.rand(1024, 1000, generator=g) * 2 - 1).cuda()
Wp = torch.zeros(1024, 1024, device="cuda"); Wp[:, :1000] = W
dH = (torch.rand(M, 1024, generator=g) * 2 - 1).cuda()
def t(name, fn, flops, n=7):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); print("%-34s %.3f ms  %.1f TF" % (name, ts[len(ts)//2], flops/ts[len(ts)//2]/1e9), flush=True)
fl = 2.0 * M * 1024 * 1000
for rep in range(2):
    t("fwd   A ld=1000, B ld=1000", lambda: ops.gemm(Y, W, relu=True), fl)
    t("fwd   A ld=1024, B ld=1024 (K=1000)", lambda: ops.gemm(Yp[:, :1000], Wp[:, :1000], relu=True), fl)
    t("fwd   A ld=1024, B ld=1024 (K=1024)", lambda: ops.gemm(Yp, Wp, relu=True), fl)
    t("dgrad out ld=1000", lambda: ops.gemm(dH, W, tb=True), fl)
    t("wgrad B ld=1000", lambda: ops.gemm(dH, Y, ta=True, tb=True), fl)
    t("wgrad B ld=1024", lambda: ops.gemm(dH, Yp[:, :1000], ta=True, tb=True), fl)
