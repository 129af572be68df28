"""Per-dispatch means of every counter rocprofv3 collected for the kernels whose name contains SUBSTR (default "n80"):
    python tools/pmc_kernel_counters.py <dir with *counter_collection.csv below it> [SUBSTR]"""
import csv, sys, glob, collections
d = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else "n80"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"]:
            agg[r["Counter_Name"]][r["Dispatch_Id"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    vals = [sum(x) for x in v.values()]
    print("%-28s n=%d  mean %.4g  min %.4g" % (k, len(vals), sum(vals) / len(vals), min(vals)))
