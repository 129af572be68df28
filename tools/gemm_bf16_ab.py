"""Same-process A/B of the two large-tile bf16 GEMM kernels (lockstep r01 vs ping-pong r02) on the hot shapes,
interleaved rounds, random operands; also checks that both give the same result.
    python tools/gemm_bf16_ab.py [--rounds 7] [--var gemm_bf16_persist --variants 0,1]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vqa_amd
ops = vqa_amd.ops
vqa_amd.lib.load()
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--variants", default="0,1")
ap.add_argument("--var", default="gemm_bf16_loop", help="library option to flip (ops.OPTIONS, e.g. gemm_bf16_persist)")
args = ap.parse_args()
variants = args.variants.split(",")
shapes = [("img fwd   (0,0)", 0, 0, 100352, 5000, 2048, False), ("img fwd bf16-out", 0, 0, 100352, 5000, 2048, True),
          ("img wgrad (1,1)", 1, 1, 5000, 2048, 100352, False), ("coatt fwd (0,0)", 0, 0, 100352, 1024, 1024, False),
          ("coatt dgrad(0,1)", 0, 1, 100352, 1024, 1024, False), ("square 8192", 0, 0, 8192, 8192, 8192, False)]
for name, ta, tb, M, N, K, ob in shapes:
    A = (torch.rand((K, M) if ta else (M, K), device="cuda") * 2 - 1).to(torch.bfloat16)
    B = (torch.rand((K, N) if tb else (N, K), device="cuda") * 2 - 1).to(torch.bfloat16)
    out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16 if ob else torch.float32)
    res, times = {}, {v: [] for v in variants}
    for v in variants:
        ops.set_option(args.var, int(v))
        ops.gemm_bf16(A, B, ta=bool(ta), tb=bool(tb), out=out, out_bf16=ob)
        torch.cuda.synchronize()
        res[v] = out.clone()
    for r in range(args.rounds):
        for v in variants:
            ops.set_option(args.var, int(v))
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(3):
                ops.gemm_bf16(A, B, ta=bool(ta), tb=bool(tb), out=out, out_bf16=ob)
            b.record(); torch.cuda.synchronize()
            times[v].append(a.elapsed_time(b) / 3)
    same = all(torch.equal(res[variants[0]], res[v]) for v in variants)
    line = "%-18s M=%6d N=%5d K=%6d" % (name, M, N, K)
    for v in variants:
        t = sorted(times[v]); med = t[len(t) // 2]
        line += " | %s=%s %.3f ms (min %.3f) %5.0f TF" % (args.var.replace("gemm_bf16_", ""), v, med, t[0], 2.0 * M * N * K / med / 1e9)
    print(line + " | identical=%s" % same, flush=True)
    del A, B, out, res
