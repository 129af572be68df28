run() { python bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$*', '|', d['ms_per_step'], '| fwd', r['frac'], r['avg_launch_ms'], '| wgrad', r['wgrad']['frac'], r['wgrad']['avg_launch_ms'])"; }
run; run --no-defer; run; run --no-defer; run --steps 60; run --no-defer --steps 60
