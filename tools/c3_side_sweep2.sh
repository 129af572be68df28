#!/bin/bash
run() { python bench.py --model mhb_coAtt --dtype bf16 --no-cpu-baseline --steps 12 --warmup 4 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline'] or {}; k=d['kernels_ms_per_step']
print('   ms_per_step %.3f  img fwd %.3f ms  wgrad %.3f ms | census: lstm fwd %.3f bwd %.3f' % (d['ms_per_step'], r.get('avg_launch_ms',0), (r.get('wgrad') or {}).get('avg_launch_ms',0), k['lstm_seq_fwd(all steps)']['ms_per_step'], k['lstm_seq_bwd(all steps)']['ms_per_step']))"; }
echo "one stream, fused node"; run
for lim in 144 128 112 96 80 64 48; do
  echo "side stream, GEMMs on <= $lim CUs"; run --overlap --side-bf16 --side-cu-limit $lim
done
echo "side stream, no limit, one workgroup per tile (VQF_GEMM_BF16_PERSIST=0)"; VQF_GEMM_BF16_PERSIST=0 run --overlap --side-bf16
echo "one stream, fused node, again"; run
