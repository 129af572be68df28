#!/bin/bash
# BASELINE config 3 (MHBCoAtt bf16, B=512): the image projection + its weight gradient on a second stream beside the
# 512-step LSTM recursion, with the persistent GEMMs confined to LIMIT CUs (library option gemm_cu_limit).
run() { python bench.py --model mhb_coAtt --dtype bf16 --no-cpu-baseline --steps 12 --warmup 4 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline'] or {}
print('   ms_per_step %.3f  img fwd %.3f ms  wgrad %.3f ms | %s' % (d['ms_per_step'], r.get('avg_launch_ms',0), (r.get('wgrad') or {}).get('avg_launch_ms',0), d['config']['streams'][:60]))"; }
echo "one stream, fused node (round 2 default)"; run
echo "one stream, fused node, again"; run
for lim in 0 248 240 224 208 192 160 128; do
  echo "side stream, bf16 projection on it, GEMMs on <= $lim CUs"; run --overlap --side-bf16 --side-cu-limit $lim
done
