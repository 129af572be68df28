"""Split-K + slab reduce vs one unsplit launch for the small products of a HieCoAtten / MFB step (M = 256 .. 7168), on live
random operands, interleaved in one process.  python tools/small_gemm_splitk_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vqa_amd
ops = vqa_amd.ops
vqa_amd.lib.load()
g = torch.Generator().manual_seed(0)
R = lambda *s: torch.randn(s, generator=g).cuda()
# (name, ta, tb, A, B, accumulate)
CASES = [
    ("CQ fwd 3584x1024x512", False, False, R(3584, 512), R(1024, 512)),
    ("dque dgrad 3584x512x1024", False, True, R(3584, 1024), R(1024, 512)),
    ("dWq2 wgrad 1024x512x3584", True, True, R(3584, 1024), R(3584, 512)),
    ("cls fwd 256x3000x1024", False, False, R(256, 1024), R(3000, 1024)),
    ("cls dgrad 256x1024x3000", False, True, R(256, 3000), R(3000, 1024)),
    ("cls wgrad 3000x1024x256", True, True, R(256, 3000), R(256, 1024)),
    ("mfb proj 512x5000x2048", False, False, R(512, 2048), R(5000, 2048)),
    ("mfb proj dgrad 512x2048x5000", False, True, R(512, 5000), R(5000, 2048)),
    ("mfb proj wgrad 5000x2048x512", True, True, R(512, 5000), R(512, 2048)),
    ("lstm x-proj 7168x4096x300", False, False, R(7168, 300), R(4096, 300)),
    ("lstm wgrad 4096x1024x7168", True, True, R(7168, 4096), R(7168, 1024)),
]


def t(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, ta, tb, A, B in CASES:
    res = []
    for rnd in range(2):
        a = t(lambda: ops.gemm(A, B, ta=ta, tb=tb))
        b = t(lambda: ops.gemm(A, B, ta=ta, tb=tb, splitk=False))
        res.append((a, b))
    print("%-32s default %6.1f / %6.1f us   unsplit %6.1f / %6.1f us" % (name, res[0][0], res[1][0], res[0][1], res[1][1]), flush=True)
