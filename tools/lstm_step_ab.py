import os, sys
sys.path.insert(0, "/root/repo")
import torch, vqa_amd
ops = vqa_amd.ops
vqa_amd.lib.load()
B, H = 512, 1024
h = torch.randn(B, H, device="cuda"); w = torch.randn(4 * H, H, device="cuda") * 0.05
pre = torch.randn(B, 4 * H, device="cuda"); cp = torch.randn(B, H, device="cuda")
c1, h1 = torch.empty(B, H, device="cuda"), torch.empty(B, H, device="cuda")
def t(fn, reps=60):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
g = pre.clone()
for rnd in range(2):
    print("gate-tile kernel %.1f us" % t(lambda: ops.lstm_step_fwd(h, w, g, cp, c1, h1)))
    with ops.options(gemm_f32_n80=0):
        print("per-wave form    %.1f us" % t(lambda: ops.lstm_step_fwd(h, w, g, cp, c1, h1)))
