#!/bin/bash
# One bench.py line per BASELINE config / mode (the table of DESIGN section 6); run on the GPU box from the repo root.
run() { timeout -k 10 300 python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '|', d['metric'], d['value'], d['ms_per_step'])"; }
run --model mhb_coAtt --dtype bf16
run --model mhb_coAtt --dtype bf16-all
run --model mhb_coAtt
run --model mfb --dtype bf16
run --model mfb --dtype bf16-all
run --model hieCoAtten
run --pruned
run --forward-only --batch 32
