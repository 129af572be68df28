"""Print the per-kernel census table of the last bench.py run (gpurun_out/bench_census.json)."""
import json
import os
import signal
import sys

signal.signal(signal.SIGPIPE, signal.SIG_DFL)

path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "bench_census.json")
d = json.load(open(path))
which = sys.argv[2] if len(sys.argv) > 2 else None
if which:
    d = d["secondary"][which]
print("ms_per_step %.3f  value %.1f" % (d["ms_per_step"], d["value"]))
tot = n = 0
for k, v in d["kernels_ms_per_step"].items():
    print("%-28s %6.1f  %.4f" % (k, v["launches_per_step"], v["ms_per_step"]))
    tot += v["ms_per_step"]
    n += v["launches_per_step"]
print("sum %.3f ms in %.0f launches" % (tot, n))
