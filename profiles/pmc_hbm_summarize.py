#!/usr/bin/env python3
"""HBM-counter summary of the HBM-bound kernels (rocprofv3 --pmc passes of tools/hbm_kernels_one.py, tools/profile_r03.sh hbm).

    python profiles/pmc_hbm_summarize.py gpurun_out/prof_r03/pmc_hbm profiles/r03_pmc_fuse [--label "..."]
writes <out>.json ({kernel: {traffic_bytes, fetch_bytes, write_bytes, algorithmic_bytes, ...}}; read by bench.py) and
<out>.txt (a table).  traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: on gfx950 FETCH_SIZE reports half of the bytes of a
16-B-per-lane streaming read, WRITE_SIZE reads exactly (MI355X_MICROARCH.md, HBM); Infinity-Cache hits are included.
Algorithmic bytes are those of the headline shapes (N=512, L=196, O=1000, D=2048, hidden 1024), as in bench.py."""
import csv
import json
import re
import sys
from collections import defaultdict

N, L, O, D, H = 512, 196, 1000, 2048, 1024
ROWS = N * L
ALG = {   # bytes one launch must move at the shapes of tools/hbm_kernels_one.py
    "mfb_fuse_fwd_kernel": 4.0 * (ROWS * 5 * O + N * 5 * O + ROWS * O + ROWS),
    "mfb_fuse_bwd_kernel": 4.0 * (2 * ROWS * 5 * O + 2 * ROWS * O + 2 * N * 4 * 5 * O),
    "scale_rows_kernel": 4.0 * 2 * ROWS * O,
    "rowdot_kernel": 4.0 * 2 * ROWS * O,
    "glimpse_pool_fwd_kernel": 4.0 * (ROWS * D + ROWS * 2 + N * 2 * D),
    "glimpse_pool_bwd_kernel": 4.0 * (ROWS * D + N * 2 * D + ROWS * 2),
    "att_logits_fwd_kernel": 4.0 * (ROWS * H + ROWS * 2),
    "att_logits_bwd_kernel": 4.0 * (2 * ROWS * H + ROWS * 2),
}
# csrc/hie.hip at BASELINE config 4's shapes (N = 256, L = 196, E = 512, T = 14; tools/hie_kernels_one.py): one (N*L, E) tensor in,
# one out (mode 2 = rank_add reads and rewrites the same one), the per-sample (T, E) / (T, L) operands are noise
HROWS, HE = 256 * 196, 512
for _m in range(4):
    ALG["hie_stream_kernel<%d>" % _m] = 4.0 * 2 * HROWS * HE
# round 5: the affinity products (vqf_hie_affinity): one pass over one (epi 1, forward) or two (epi 2, gradient) (N*L, E) tensors
ALG["hie_affinity_kernel<1>"] = 4.0 * HROWS * HE
ALG["hie_affinity_kernel<2>"] = 4.0 * 2 * HROWS * HE
BENCH_NAME = {"hie_affinity_kernel<1>": "hie_affinity", "hie_affinity_kernel<2>": "hie_affinity", "hie_stream_kernel<0>": "hie_hv_fwd", "hie_stream_kernel<1>": "hie_head_bwd", "hie_stream_kernel<2>": "hie_rank_add",
              "hie_stream_kernel<3>": "hie_rank_left","mfb_fuse_fwd_kernel": "mfb_fuse_fwd", "mfb_fuse_bwd_kernel": "mfb_fuse_bwd", "scale_rows_kernel": "scale_rows",
              "rowdot_kernel": "rowdot", "glimpse_pool_fwd_kernel": "glimpse_pool_fwd", "glimpse_pool_bwd_kernel": "glimpse_pool_bwd",
              "att_logits_fwd_kernel": "att_logits_fwd", "att_logits_bwd_kernel": "att_logits_bwd"}


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"hie_stream_kernel<(\d)", name)           # hie_stream_kernel<MODE[, TMAX]>
    if m:
        return "hie_stream_kernel<%s>" % m.group(1)
    m = re.match(r"hie_affinity_kernel<(\d)", name)
    if m:
        return "hie_affinity_kernel<%s>" % m.group(1)
    m = re.match(r"([A-Za-z0-9_]+)", name)
    return m.group(1) if m else name


def main(src, out, label):
    vals = defaultdict(lambda: defaultdict(list))
    full = {}
    scratch = {}
    for p in ("p1", "p2", "p3"):
        for r in csv.DictReader(open("%s/%s_counter_collection.csv" % (src, p))):
            k = short(r["Kernel_Name"])
            if k not in ALG:
                continue
            vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            full[k] = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]).split("(")[0]
            scratch[k] = int(r.get("Scratch_Size", 0) or 0)
    dur = defaultdict(list)
    for r in csv.DictReader(open("%s/p2_kernel_trace.csv" % src)):
        k = short(r["Kernel_Name"])
        if k in ALG:
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    mean = lambda v: sum(v) / len(v) if v else 0.0
    res = {}
    for k in ALG:
        c = vals.get(k)
        if not c:
            continue
        fetch, write = 2.0 * mean(c["FETCH_SIZE"]) * 1024, mean(c["WRITE_SIZE"]) * 1024
        hit, miss = mean(c["TCC_HIT_sum"]), mean(c["TCC_MISS_sum"])
        ms = sorted(dur[k])[len(dur[k]) // 2] if dur[k] else 0.0
        res[BENCH_NAME[k]] = {
            "kernel": full[k], "fetch_bytes": fetch, "write_bytes": write, "traffic_bytes": fetch + write,
            "algorithmic_bytes": ALG[k], "traffic_over_algorithmic": round((fetch + write) / ALG[k], 3),
            "l2_hit_rate": round(hit / max(hit + miss, 1.0), 3), "scratch_bytes_per_lane": scratch.get(k, 0),
            "ms_profiled": round(ms, 4), "hbm_gbs_profiled": round((fetch + write) / (ms * 1e-3) / 1e9, 1) if ms else None,
            "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024  [gfx950: FETCH_SIZE counts half of 16-B/lane streaming reads]",
            "launches": len(c["FETCH_SIZE"]), "source": out + ".txt"}
    json.dump(res, open(out + ".json", "w"), indent=1)
    with open(out + ".txt", "w") as f:
        f.write("# %s\n\n" % label)
        f.write("`tools/profile_r0X.sh hbm|hie`: three separate `rocprofv3 --kernel-trace --pmc <group> -- python3 tools/hbm_kernels_one.py | hie_kernels_one.py` passes\n"
                "(p2 = FETCH_SIZE ..., p3 = WRITE_SIZE TCC_HIT_sum TCC_MISS_sum), 3 launches per kernel at the headline shapes, means per dispatch.\n"
                "traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 correction; Infinity-Cache hits included); ms = median of the p2 pass.\n\n")
        f.write("| kernel | fetch GB | write GB | traffic GB | algorithmic GB | traffic / alg | L2 hit | scratch B/lane | ms (profiled) | GB/s (counter bytes) |\n|---|---|---|---|---|---|---|---|---|---|\n")
        for k, r in res.items():
            f.write("| `%s` | %.3f | %.3f | %.3f | %.3f | **%.2f** | %.1f %% | %d | %.4f | %s |\n" % (
                r["kernel"], r["fetch_bytes"] / 1e9, r["write_bytes"] / 1e9, r["traffic_bytes"] / 1e9, r["algorithmic_bytes"] / 1e9,
                r["traffic_over_algorithmic"], 100 * r["l2_hit_rate"], r["scratch_bytes_per_lane"], r["ms_profiled"], r["hbm_gbs_profiled"]))
    print(open(out + ".txt").read())


if __name__ == "__main__":
    lab = "HBM counters of the HBM-bound kernels (MI355X, ROCm 7.2)"
    if "--label" in sys.argv:
        i = sys.argv.index("--label")
        lab = sys.argv[i + 1]
        del sys.argv[i:i + 2]
    main(sys.argv[1], sys.argv[2], lab)
