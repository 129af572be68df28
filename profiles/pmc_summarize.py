#!/usr/bin/env python3
"""Per-kernel PMC summary from rocprofv3 --pmc CSVs (counter_collection + kernel_trace of the same pass).

    python profiles/pmc_summarize.py gpurun_out/pmc_r01/p1 gpurun_out/pmc_r01/p2 ...
Prints, for each kernel whose name matches FILTER (default 'gemm_f32_kernel'), the mean counter value
per dispatch, grouped by grid size.
"""
import csv
import re
import sys
from collections import defaultdict

import os
FILTER = os.environ.get("PMC_FILTER", "gemm_f32_kernel")


def main(prefixes):
    out = defaultdict(lambda: defaultdict(list))
    for pre in prefixes:
        for r in csv.DictReader(open(pre + "_counter_collection.csv")):
            name = r["Kernel_Name"]
            if FILTER not in name:
                continue
            name = re.sub(r"\(anonymous namespace\)::", "", name)
            name = re.sub(r"\(.*", "", name).replace("void ", "")
            key = (name, r.get("Grid_Size", r.get("Grid_Size_X", "?")))
            out[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key, cs in sorted(out.items()):
        print("## %s grid %s" % key)
        for c, v in sorted(cs.items()):
            print("  %-32s mean %.6g  (n=%d)" % (c, sum(v) / len(v), len(v)))


if __name__ == "__main__":
    main(sys.argv[1:])
