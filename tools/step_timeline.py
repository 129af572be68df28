"""One training step out of a rocprofv3 --kernel-trace CSV: every kernel in start order with its duration and the idle gap in
front of it, the step delimited by two consecutive launches of `--mark` (default adam_kernel).  Usage:
    python tools/step_timeline.py <kernel_trace.csv> [--mark adam_kernel] [--step -2] [--min-us 0]"""
import argparse, csv, re, collections
ap = argparse.ArgumentParser()
ap.add_argument("csv"); ap.add_argument("--mark", default="adam_kernel"); ap.add_argument("--step", type=int, default=-2)
ap.add_argument("--min-us", type=float, default=0.0)
a = ap.parse_args()
rows = list(csv.DictReader(open(a.csv)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if a.mark in r["Kernel_Name"]]
lo, hi = marks[a.step - 1] + 1, marks[a.step] + 1
sel = rows[lo:hi]
t0 = int(rows[lo - 1]["End_Timestamp"])
prev = t0
busy = 0
agg = collections.OrderedDict()
print("step of %d kernels, %.1f us from the end of the previous %s to the end of this one" % (len(sel), (int(sel[-1]["End_Timestamp"]) - t0) / 1e3, a.mark))
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name)[:70]
    busy += e - s
    d = agg.setdefault(name, [0, 0.0, 0.0])
    d[0] += 1; d[1] += (e - s) / 1e3; d[2] += max(0, s - prev) / 1e3
    if (e - s) / 1e3 >= a.min_us:
        print("%9.1f  +%6.1f gap  %8.1f us  grid %-8s %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, r.get("Grid_Size", "?"), name))
    prev = max(prev, e)
print("busy %.1f us, idle %.1f us" % (busy / 1e3, (int(sel[-1]["End_Timestamp"]) - t0 - busy) / 1e3))
print("\nby kernel (launches, total us, total gap in front us):")
for k, (n, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-70s %3d  %8.1f  %7.1f" % (k, n, d, g))
