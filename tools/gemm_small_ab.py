"""Small-M fp32 products (the LSTM's recurrent GEMMs and the M = 512 projections): gemm_f32_wave.hip on / off
(library option gemm_f32_wave), same process, interleaved rounds, incl. the split-K reduce launch of the old path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vqa_amd
ops = vqa_amd.ops
vqa_amd.lib.load()
SH = [("lstm fwd  acc", 0, 512, 4096, 1024, True), ("lstm bwd", 1, 512, 1024, 4096, False),
      ("proj 5000x2048", 0, 512, 5000, 2048, False), ("dgrad 2048x5000", 1, 512, 2048, 5000, False),
      ("lstm fwd M=256", 0, 256, 4096, 1024, True)]
for name, tb, M, N, K, acc in SH:
    A = (torch.rand((M, K), device="cuda") - 0.5)
    B = (torch.rand((K, N) if tb else (N, K), device="cuda") - 0.5) * 0.1
    out = torch.zeros((M, N), device="cuda")
    t = {"0": [], "1": [], "2": [], "3": []}
    for v in ("0", "1", "2", "3"):
        ops.set_option("gemm_f32_wave", int(v))
        ops.gemm(A, B, tb=bool(tb), out=out, accumulate=acc)
    torch.cuda.synchronize()
    for r in range(7):
        for v in ("0", "1", "2", "3"):
            ops.set_option("gemm_f32_wave", int(v))
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                ops.gemm(A, B, tb=bool(tb), out=out, accumulate=acc)
            b.record(); torch.cuda.synchronize()
            t[v].append(a.elapsed_time(b) / 20 * 1e3)
    m0, m1, m2, m3 = sorted(t["0"])[3], sorted(t["1"])[3], sorted(t["2"])[3], sorted(t["3"])[3]
    fl = 2.0 * M * N * K
    print("%-16s (0,%d) M=%4d N=%5d K=%5d | 128x128 + reduce %.1f us %.0f TF | wave, private B %.1f us %.0f TF | wave, B once per workgroup %.1f us %.0f TF | ... 12 slots %.1f us %.0f TF" % (
        name, tb, M, N, K, m0, fl / m0 / 1e6, m1, fl / m1 / 1e6, m2, fl / m2 / 1e6, m3, fl / m3 / 1e6), flush=True)
