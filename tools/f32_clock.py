"""Clock and per-tile time budget of the large-tile fp32 GEMM, from a diagnostic build that stamps s_memtime /
s_memrealtime at kernel entry, K-loop start, K-loop end and after the tile's stores have landed (never the product
build, never quote its run time; MI355X_MICROARCH.md, DVFS give-back item 6):
    tools/build_variant.sh clock "-DVQF_F32BIG_CLOCK" gemm_f32_big.hip
    VQF_LIB=variants/libvqf_clock.so python tools/f32_clock.py
in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz over entry -> K-loop end, after >= 2 s of back-to-back
launches on random data; a slab is MFMA-bound at 8192 cycles per SIMD (2 waves x 64 MFMAs x 64 cycles)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, vqa_amd
ops = vqa_amd.ops
vqa_amd.lib.load()
for name, ta, tb, M, N, K in (("img fwd (0,0)", 0, 0, 100352, 5000, 2048), ("sq 8192 (0,0)", 0, 0, 8192, 8192, 8192),
                              ("sq 8192 (1,1)", 1, 1, 8192, 8192, 8192)):
    A = torch.rand((K, M) if ta else (M, K), device="cuda") * 2 - 1
    B = torch.rand((K, N) if tb else (N, K), device="cuda") * 2 - 1
    out = torch.empty((M, N), device="cuda")
    t0 = time.time()
    n = 0
    while time.time() - t0 < 2.5:                      # the clock settles under sustained load
        for _ in range(10):
            ops.gemm(A, B, ta=bool(ta), tb=bool(tb), out=out)
        torch.cuda.synchronize()
        n += 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.gemm(A, B, ta=bool(ta), tb=bool(tb), out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    ws = ops.workspace(A.device, ops.SPLITK_WS_BYTES)
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    d = ws[:tiles * 2 * 8 * 8].view(torch.int64).view(tiles, 2, 8).cpu().numpy().astype(np.float64)
    clk = np.median(d[:, :, 4] / d[:, :, 3]) * 0.1      # GHz
    S = d[0, 0, 5]
    pro, loop, epi = (np.median(d[:, :, i]) for i in range(3))
    tf = 2.0 * M * N * K / ms / 1e9
    peak_at_clk = 157.3 * clk / 2.4
    print("%-14s %.3f ms %.1f TF  in-kernel clock %.3f GHz (fp32 MFMA peak at that clock %.1f TF -> %.3f of it)  per tile: "
          "prologue %.0f cyc, K loop %.0f cyc = %.0f per slab (8192 = MFMA-bound, %.3f), epilogue incl. landed stores %.0f cyc"
          % (name, ms, tf, clk, peak_at_clk, tf / peak_at_clk, pro, loop, loop / S, 8192.0 / (loop / S), epi), flush=True)
    # time-line per persistent workgroup (work items w, w + 256, ...): its span on the 100 MHz clock, and the spread
    # of the tile times and clocks over the chip -- the launch ends with its slowest workgroup
    lp = d[:, 0, 1]
    print("               tiles %d  K-loop cycles per tile: p5 %.0f median %.0f p95 %.0f max %.0f   clock per tile: min %.3f "
          "median %.3f max %.3f GHz" % (tiles, np.percentile(lp, 5), np.median(lp), np.percentile(lp, 95), lp.max(),
                                        (d[:, 0, 4] / d[:, 0, 3]).min() * 0.1, clk, (d[:, 0, 4] / d[:, 0, 3]).max() * 0.1))
    nwg = min(tiles, 256)
    r0 = d[:, 0, 6]
    dur = d[:, 0, 3] + d[:, 0, 2] / (d[:, 0, 4] / d[:, 0, 3])          # entry -> stores landed, 100 MHz ticks
    spans, sums = [], []
    for b in range(nwg):
        idx = np.arange(b, tiles, nwg)
        spans.append((r0[idx[-1]] + dur[idx[-1]] - r0[idx[0]]) * 0.01)   # us
        sums.append(dur[idx].sum() * 0.01)
    spans, sums = np.array(spans), np.array(sums)
    first = r0[:nwg]
    print("               per workgroup: span min %.1f median %.1f max %.1f us (launch %.1f us); sum of its tiles' times max %.1f us; "
          "first-tile start spread %.1f us; span by XCD (median): %s"
          % (spans.min(), np.median(spans), spans.max(), ms * 1e3, sums.max(), (first.max() - first.min()) * 0.01,
             " ".join("%.0f" % np.median(spans[x::8]) for x in range(8))), flush=True)
