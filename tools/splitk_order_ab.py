"""Same-process A/B of the split-K work-item order (library option gemm_splitk_order; csrc/gemm_bf16_big.hip tile_coord):
the image projection's weight gradient dW = dP^T X (M=5000, N=2048, K=100352), fp32 and bf16, live operands; interleaved
rounds, hipEvent-timed, results compared bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vqa_amd
ops = vqa_amd.ops
vqa_amd.lib.load()
R = 512 * 196
g = torch.Generator().manual_seed(0)
A32 = ((torch.rand((R, 5000), generator=g) - 0.5) * 0.1).cuda()
B32 = torch.relu(torch.randn((R, 2048), generator=g)).cuda()
for dtype in ("f32", "bf16"):
    if dtype == "bf16":
        A, B = A32.to(torch.bfloat16), B32.to(torch.bfloat16)
        fn = lambda: ops.gemm_bf16(A, B, ta=True, tb=True)
    else:
        A, B = A32, B32
        fn = lambda: ops.gemm(A, B, ta=True, tb=True)
    res, ms = {}, {0: [], 1: []}
    for rnd in range(4):
        for opt in (0, 1):
            with ops.options(gemm_splitk_order=opt):
                fn()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(5):
                    out = fn()
                b.record()
                torch.cuda.synchronize()
                ms[opt].append(a.elapsed_time(b) / 5)
                res[opt] = out.clone()
    same = torch.equal(res[0], res[1])
    fl = 2.0 * 5000 * 2048 * R
    for opt in (0, 1):
        best = min(ms[opt])
        print("%s wgrad order=%d: ms per launch %s  best %.3f = %.1f TF" % (dtype, opt, ["%.3f" % m for m in ms[opt]], best, fl / best / 1e9))
    print("%s: results bit-identical across orders: %s" % (dtype, same))
    assert same
