"""Same-process A/B of the fp32 GEMM kernels on the hot shapes (random, LIVE operands), interleaved rounds:
  small = 128x128 kernel (library option gemm_f32_big=0), lock = 256x256 lockstep loop, pp = 256x256 ping-pong loop,
  stag = lockstep loop with waves 4-7 half a slab behind.          python tools/gemm_f32_ab.py [--rounds 5] [--shapes fwd,wgrad]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vqa_amd
ops = vqa_amd.ops
vqa_amd.lib.load()
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--shapes", default="fwd,wgrad,coatt_fwd,coatt_dgrad,sq")
args = ap.parse_args()
ENV = {"small": {"gemm_f32_big": 0, "gemm_f32_loop": 1}, "lock": {"gemm_f32_big": 1, "gemm_f32_loop": 0},
       "pp": {"gemm_f32_big": 1, "gemm_f32_loop": 1}, "stag": {"gemm_f32_big": 1, "gemm_f32_loop": 2}}


def _set(e):
    for k, v in e.items():
        ops.set_option(k, v)
SH = {"fwd": (0, 0, 100352, 5000, 2048), "wgrad": (1, 1, 5000, 2048, 100352), "coatt_fwd": (0, 0, 100352, 1024, 1024),
      "coatt_dgrad": (0, 1, 100352, 1024, 1024), "sq": (0, 0, 8192, 8192, 8192), "sq_tn": (1, 0, 8192, 8192, 8192),
      "coatt_wgrad": (1, 1, 1024, 1000, 100352), "coatt1000": (0, 0, 100352, 1024, 1000)}
for name in args.shapes.split(","):
    ta, tb, M, N, K = SH[name]
    g = torch.Generator(device="cpu").manual_seed(1)
    A = torch.relu(torch.randn((K, M) if ta else (M, K), generator=g)).cuda() if name == "fwd" else \
        ((torch.rand((K, M) if ta else (M, K), generator=g) - 0.5) * 0.1).cuda()
    B = (torch.relu(torch.randn((K, N) if tb else (N, K), generator=g)) if name == "wgrad" else
         (torch.rand((K, N) if tb else (N, K), generator=g) - 0.5) * 0.06).cuda()
    out = torch.empty((M, N), device="cuda")
    res, times = {}, {v: [] for v in ENV}
    for v, e in ENV.items():
        _set(e)
        ops.gemm(A, B, ta=bool(ta), tb=bool(tb), out=out)
        torch.cuda.synchronize()
        res[v] = out.clone()
    for r in range(args.rounds):
        for v, e in ENV.items():
            _set(e)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(2):
                ops.gemm(A, B, ta=bool(ta), tb=bool(tb), out=out)
            b.record(); torch.cuda.synchronize()
            times[v].append(a.elapsed_time(b) / 2)
    ref = res["small"].double()
    line = "%-12s (%d,%d) M=%6d N=%5d K=%6d" % (name, ta, tb, M, N, K)
    for v in ENV:
        t = sorted(times[v]); med = t[len(t) // 2]
        d = float((res[v].double() - ref).abs().max() / ref.abs().max())
        line += " | %s %.3f ms (min %.3f) %5.1f TF d=%.1e" % (v, med, t[0], 2.0 * M * N * K / med / 1e9, d)
    print(line, flush=True)
    del A, B, out, res
