"""Tile-quantisation probe for the 128x128 fp32 kernel (gemm_f32.hip, 4 workgroups per CU = 1024 slots): the mid-size GEMMs of
the step at their own M and at neighbouring M whose tile count is a whole number of rounds.  If TF/s jumps at the whole-round
sizes, the last partial round is what these shapes lose; if not, it is the per-tile prologue / epilogue."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vqa_amd
ops = vqa_amd.ops
vqa_amd.lib.load()
g = torch.Generator(device="cuda").manual_seed(0)


def run(tag, M, N, K, ta=False, tb=False, reps=8):
    A = torch.randn((K, M) if ta else (M, K), device="cuda", generator=g)
    B = torch.randn((K, N) if tb else (N, K), device="cuda", generator=g)
    for _ in range(2):
        ops.gemm(A, B, ta, tb)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        ops.gemm(A, B, ta, tb)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    print("%-34s M=%6d N=%5d K=%5d (%d,%d)  tiles %5d = %.3f rounds of 1024 | %.4f ms  %.1f TF" % (
        tag, M, N, K, ta, tb, tiles, tiles / 1024.0, ms, 2.0 * M * N * K / ms / 1e9), flush=True)


for M in (100352, 98304, 114688, 65536, 131072):
    run("co_att_conv1 fwd", M, 1024, 1000)
for M in (100352, 98304, 114688):
    run("co_att_conv1 dgrad", M, 1000, 1024, False, True)
for M in (50176, 49152, 32768, 65536, 57344):
    run("HieCoAtten img_emb fwd", M, 512, 2048)
for M in (50176, 32768, 65536):
    run("HieCoAtten img_emb dgrad", M, 2048, 512, False, True)
for K in (1000, 1024, 2048, 4096):
    run("K sweep at 6 whole rounds", 98304, 1024, K)
