#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel AND per launch geometry (grid size),
so that the dominant GEMM launches are not averaged with the small ones.

    python profiles/summarize.py gpurun_out/prof_r01/r01_kernel_trace.csv > profiles/r01_kernel_summary.md
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:<>, ]+?)\(", name)
    s = (m.group(1) if m else name).strip()
    return s[:70]


def main(path):
    rows = list(csv.DictReader(open(path)))
    groups = defaultdict(list)
    for r in rows:
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        grid = "%sx%sx%s" % (r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?"))
        wg = r.get("Workgroup_Size_X", "?")
        key = (short(r["Kernel_Name"]), grid, wg, r.get("VGPR_Count", "?"), r.get("LDS_Block_Size", "?"))
        groups[key].append(dur)
    # persistent kernels launch one workgroup per CU whatever the shape, so one (kernel, grid) may hold several shapes:
    # split a group where consecutive sorted durations jump by more than 3x
    split = {}
    for key, v in groups.items():
        v = sorted(v)
        parts, cur = [], [v[0]]
        for d in v[1:]:
            if d > 3 * cur[-1]:
                parts.append(cur)
                cur = []
            cur.append(d)
        parts.append(cur)
        for i, part in enumerate(parts):
            k = key if len(parts) == 1 else (key[0] + " [duration cluster %d of %d]" % (i + 1, len(parts)),) + key[1:]
            split[k] = part
    groups = split
    total = sum(sum(v) for v in groups.values())
    print("| kernel | grid (threads) | wg | VGPR | LDS B | calls | avg ms | min ms | max ms | total ms | % |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for key, v in sorted(groups.items(), key=lambda kv: -sum(kv[1]))[:40]:
        k, grid, wg, vg, lds = key
        print("| `%s` | %s | %s | %s | %s | %d | %.4f | %.4f | %.4f | %.3f | %.2f |" % (
            k, grid, wg, vg, lds, len(v), sum(v) / len(v) / 1e6, min(v) / 1e6, max(v) / 1e6, sum(v) / 1e6,
            100.0 * sum(v) / total))
    print("\ntotal kernel time in trace: %.3f ms over %d dispatches" % (total / 1e6, len(rows)))


if __name__ == "__main__":
    main(sys.argv[1])
