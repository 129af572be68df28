#!/bin/bash
# beyond-L2 traffic of the weight-gradient launches (fp32 and bf16) with the adaptive tile-group size: FETCH_SIZE and
# WRITE_SIZE / L2 hit counters in two separate rocprofv3 --pmc passes each.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_gm2; mkdir -p $O
for v in f32 bf16; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/$v -o p2 -- python3 $R/tools/gemm_one.py --dtype $v --shape wgrad > $O/$v.p2.log 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/$v -o p3 -- python3 $R/tools/gemm_one.py --dtype $v --shape wgrad > $O/$v.p3.log 2>&1
done
ls $O/*
