#!/bin/bash
# One bench.py line per BASELINE config / mode (the table of DESIGN section 6); run on the GPU box from the repo root.
run() { timeout -k 10 300 python bench.py "$@" --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}
print('$*', '|', d['metric'], d['value'], d['ms_per_step'], '| dominant GEMM frac', r.get('frac'), r.get('avg_launch_ms'))"; }
run
run --steps 300 --warmup 5
run --overlap
run --no-overlap
run --model mhb_coAtt --dtype bf16
run --model mhb_coAtt --dtype bf16 --overlap --side-bf16 --side-cu-limit 128
run --model mhb_coAtt --dtype bf16-all
run --model mhb_coAtt --dtype bf16-all --overlap --side-bf16 --side-cu-limit 128
run --model mhb_coAtt
run --model mhb_coAtt --overlap --side-cu-limit 128
run --model mfb --dtype bf16
run --model mfb --dtype bf16-all
run --model hieCoAtten
run --pruned
run --forward-only --batch 32
run --forward-only
