"""The M = 512 projection shapes of the headline step (ques_proj*, img_proj2 and their gradients, mfb.py:92,126-127) under the
kernel-routing options: which family runs them fastest.   python tools/gemm_m512_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vqa_amd
ops = vqa_amd.ops
vqa_amd.lib.load()
g = torch.Generator(device="cuda").manual_seed(3)
rn = lambda *s: torch.randn(s, device="cuda", generator=g)
shapes = [(0, 0, 512, 5000, 2048), (0, 0, 512, 5000, 4096), (0, 1, 512, 2048, 5000), (0, 1, 512, 4096, 5000),
          (1, 1, 5000, 2048, 512), (1, 1, 5000, 4096, 512), (0, 0, 512, 1000, 1000), (1, 1, 1000, 1000, 512),
          (0, 0, 7168, 1024, 1024), (0, 1, 7168, 1024, 1024), (1, 1, 1024, 1024, 7168), (0, 0, 7168, 4096, 300),
          (0, 0, 7168, 512, 1024), (0, 1, 7168, 1024, 512), (1, 1, 512, 1024, 7168), (0, 1, 7168, 300, 4096), (1, 1, 4096, 300, 7168),
          (1, 1, 4096, 1024, 7168)]
variants = [("default", {}), ("big=2", dict(gemm_f32_big=2)), ("big=0", dict(gemm_f32_big=0)), ("unfused-splitk", dict(gemm_splitk_fused=0)),
            ("wave=0", dict(gemm_f32_wave=0)), ("nosplit", None)]
print("%-26s" % "shape (ta,tb,M,N,K)" + "".join("%16s" % v[0] for v in variants))
for ta, tb, M, N, K in shapes:
    A = rn(K, M) if ta else rn(M, K)
    B = rn(K, N) if tb else rn(N, K)
    row = []
    for name, opt in variants:
        kw = {}
        if opt is None:
            opt, kw = {}, dict(splitk=False)
        with ops.options(**opt):
            for _ in range(3):
                ops.gemm(A, B, ta=bool(ta), tb=bool(tb), **kw)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            R = 20
            for _ in range(R):
                ops.gemm(A, B, ta=bool(ta), tb=bool(tb), **kw)
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / R
        row.append("%7.1f us %4.0fTF" % (ms * 1e3, 2.0 * M * N * K / ms / 1e9))
    print("%-26s" % str((ta, tb, M, N, K)) + "".join("%16s" % r for r in row))
