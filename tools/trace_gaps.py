"""Per-kernel duration and gap-to-previous statistics from a rocprofv3 kernel-trace CSV."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = r["Kernel_Name"][:60]
    dur[k].append(e - s)
    if prev_end is not None:
        gap[k].append(s - prev_end)
    prev_end = e
for k in sorted(dur, key=lambda k: -sum(dur[k]))[:12]:
    d, g = sorted(dur[k]), sorted(gap[k]) or [0]
    print("%-60s n=%5d dur med %7.2f us  p10 %7.2f p90 %7.2f | gap med %6.2f us p90 %6.2f" % (
        k, len(d), d[len(d)//2]/1e3, d[len(d)//10]/1e3, d[9*len(d)//10]/1e3, g[len(g)//2]/1e3, g[9*len(g)//10]/1e3))
