"""Mid-size fp32 GEMMs: 128x128 kernel vs the 256x256 persistent kernel forced on (library option gemm_f32_big = 2) vs the
default routing (whole rounds on the 256x256 kernel + the remaining rows on the 128x128 one), at the shapes of the step and at
neighbouring M whose 256x256 tile count is a whole number of rounds of 256 CUs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vqa_amd
ops = vqa_amd.ops
vqa_amd.lib.load()
g = torch.Generator(device="cuda").manual_seed(0)


def t(A, B, ta, tb, reps=6):
    for _ in range(2):
        ops.gemm(A, B, ta, tb)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        ops.gemm(A, B, ta, tb)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def run(tag, M, N, K, ta=False, tb=False):
    A = torch.randn((K, M) if ta else (M, K), device="cuda", generator=g)
    B = torch.randn((K, N) if tb else (N, K), device="cuda", generator=g)
    forms = (("128x128", {"gemm_f32_big": 0}), ("256x256 forced", {"gemm_f32_big": 2, "gemm_f32_rounds": 0}),
             ("default (row split)", {}))
    best = {}
    for rnd in range(2):
        for name, opts in forms:
            with ops.options(**opts):
                best[name] = min(best.get(name, 1e9), t(A, B, ta, tb))
    rows = ops.gemm_big_rows(ta, tb, M, N, K)
    t256 = ((M + 255) // 256) * ((N + 255) // 256)
    fl = 2.0 * M * N * K / 1e9
    print("%-26s M=%6d N=%5d K=%5d (%d,%d) | " % (tag, M, N, K, ta, tb) + " | ".join(
        "%s %.4f ms %.1f TF" % (n, best[n], fl / best[n]) for n, _ in forms) + " | %d tiles = %.3f rounds, %d rows on the large tiles" % (
        t256, t256 / 256.0, rows), flush=True)


run("warm-up", 32768, 1024, 1024)
for M in (100352, 98304):
    run("co_att_conv1 fwd", M, 1024, 1000)
for M in (100352, 98304):
    run("co_att_conv1 dgrad", M, 1000, 1024, False, True)
for M in (50176, 49152, 65536):
    run("HieCoAtten img_emb fwd", M, 512, 2048)
for M in (50176, 65536):
    run("HieCoAtten img_emb dgrad", M, 2048, 512, False, True)
run("HieCoAtten 512x512", 50176, 512, 512)
run("q-att conv", 7168, 1024, 1024)
run("lstm input projection", 7168, 4096, 304)
run("M=512 proj", 512, 5000, 2048)
