#!/usr/bin/env python3
"""One record per dominant GEMM launch from the three rocprofv3 --pmc passes of tools/profile_r0X.sh gemm|mid:

    python profiles/pmc_gemm_summarize.py gpurun_out/prof_r04 profiles/r04_pmc_gemm f32_fwd:f32:100352:5000:2048 f32_wgrad:f32:5000:2048:100352 ...

reads <dir>/pmc_<name>/p{1,2,3}_{counter_collection,kernel_trace}.csv, keeps the kernel with the largest total time of each pass
directory (the GEMM; a split-K reduce launch, if any, is listed beside it) and writes <out>.json (a list of records read by
bench.py::pmc_lookup) and <out>.txt (a table).  clock = GRBM_GUI_ACTIVE / 8 XCDs / dispatch time of the same pass;
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes
(gfx950: FETCH_SIZE counts half of a 16-byte-per-lane streaming read, MI355X_MICROARCH.md; Infinity-Cache hits are included);
L2 hit rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum).  Means over the profiled dispatches EXCEPT the first of each pass
(it runs before the clock has settled)."""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name).strip()


def load(d, p):
    cnt = defaultdict(lambda: defaultdict(list))       # kernel -> counter -> values in dispatch order
    for r in csv.DictReader(open("%s/p%d_counter_collection.csv" % (d, p))):
        cnt[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = defaultdict(list)
    for r in csv.DictReader(open("%s/p%d_kernel_trace.csv" % (d, p))):
        dur[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6))
    return cnt, {k: [ms for _, ms in sorted(v)] for k, v in dur.items()}


def settled(v):
    return v[1:] if len(v) > 1 else v


def mean(v):
    return sum(v) / max(len(v), 1)


def main(root, out, specs):
    recs, lines = [], []
    for spec in specs:
        name, dtype, M, N, K = spec.split(":")
        M, N, K = int(M), int(N), int(K)
        d = "%s/pmc_%s" % (root, name)
        passes = [load(d, p) for p in (1, 2, 3)]
        durs = passes[0][1]
        kern = max(durs, key=lambda k: sum(durs[k]))
        c = {}
        for cnt, _ in passes:
            for cname, vals in cnt.get(kern, {}).items():
                c[cname] = mean(settled(vals))
        ms_all = passes[0][1][kern]
        ms = mean(settled(ms_all))
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        clock = gui / 8.0 / (ms * 1e-3) / 1e9 if ms else 0.0
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * gui / 8.0) if gui else 0.0
        traffic = (2.0 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024.0
        hit, miss = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0)
        esz = 2.0 if dtype == "bf16" else 4.0
        alg = esz * (M * K + N * K) + 4.0 * M * N
        others = {k: [round(x, 4) for x in v] for k, v in durs.items() if k != kern}
        rec = {"name": name, "kernel": kern, "dtype": dtype, "M": M, "N": N, "K": K, "FETCH_SIZE_KB": c.get("FETCH_SIZE"),
               "WRITE_SIZE_KB": c.get("WRITE_SIZE"), "traffic_bytes": traffic,
               "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024  [gfx950: FETCH_SIZE counts half of 16-B/lane streaming reads]",
               "note": "bytes leaving L2 (Infinity-Cache hits included); algorithmic bytes %.3g (operands once + fp32 output)" % alg,
               "mfma_busy_frac": round(busy, 4), "l2_hit_rate": round(hit / (hit + miss), 4) if hit + miss else None,
               "clock_ghz": round(clock, 3), "ms_profiled": [round(x, 4) for x in ms_all], "other_kernels_ms": others,
               "source": out + ".txt"}
        recs.append(rec)
        lines.append("| %s M=%d N=%d K=%d | `%s` | %s | %.2f GHz | %.1f %% | %s | %.2f GB | %.2f GB |" % (
            name, M, N, K, kern, " / ".join("%.3f" % x for x in ms_all), clock, 100 * busy,
            ("%.1f %%" % (100 * rec["l2_hit_rate"])) if rec["l2_hit_rate"] is not None else "-", traffic / 1e9, alg / 1e9))
        lines.append("".join("\n    %-28s %.6g" % kv for kv in sorted(c.items())))
    json.dump(recs, open(out + ".json", "w"), indent=1)
    with open(out + ".txt", "w") as f:
        f.write("# PMC passes over the dominant GEMM launches (tools/profile_r0X.sh; profiles/pmc_gemm_summarize.py; formulas in its docstring)\n\n")
        f.write("| launch | kernel | ms per profiled dispatch | clock | MFMA busy | L2 hit | beyond-L2 traffic | algorithmic |\n|---|---|---|---|---|---|---|---|\n")
        for ln in lines:
            if ln.startswith("|"):
                f.write(ln + "\n")
        f.write("\nmean counters per dispatch (first dispatch of each pass excluded):\n")
        for rec, ln in zip(recs, [l for l in lines if not l.startswith("|")]):
            f.write("\n## %s  %s%s\n" % (rec["name"], rec["kernel"], ln))
    print(open(out + ".txt").read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
