"""Same-process A/B of the short-edge-tile path of the fp32 large-tile GEMM (library option gemm_f32_edge) on the image
projection's forward launch (M=100352, N=5000, K=2048, LIVE operands), interleaved rounds, persistent and per-tile launches."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vqa_amd
ops = vqa_amd.ops
vqa_amd.lib.load()
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=7)
args = ap.parse_args()
M, N, K = 100352, 5000, 2048
g = torch.Generator(device="cpu").manual_seed(1)
A = torch.relu(torch.randn((M, K), generator=g)).cuda()
B = ((torch.rand((N, K), generator=g) - 0.5) * 0.06).cuda()
bias = torch.zeros(N, device="cuda")
out = torch.empty((M, N), device="cuda")
VAR = {"edge0": dict(gemm_f32_edge=0), "edge1": dict(gemm_f32_edge=1),
       "edge0,per-tile": dict(gemm_f32_edge=0, gemm_f32_persist=0), "edge1,per-tile": dict(gemm_f32_edge=1, gemm_f32_persist=0)}
times = {v: [] for v in VAR}
for v, o in VAR.items():
    with ops.options(**o):
        ops.gemm(A, B, bias=bias, out=out)
torch.cuda.synchronize()
for r in range(args.rounds):
    for v, o in VAR.items():
        with ops.options(**o):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(2):
                ops.gemm(A, B, bias=bias, out=out)
            b.record(); torch.cuda.synchronize()
            times[v].append(a.elapsed_time(b) / 2)
for v in VAR:
    t = sorted(times[v]); med = t[len(t) // 2]
    print("%-16s %.3f ms (min %.3f)  %.1f TF  %.4f of 157.3" % (v, med, t[0], 2.0 * M * N * K / med / 1e9, 2.0 * M * N * K / med / 1e9 / 157.3), flush=True)
