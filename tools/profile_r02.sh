#!/bin/bash
# rocprofv3 evidence of round 2 (run on the GPU box from the repo root): kernel trace of the bench step and three
# separate PMC passes (never combined with other trace domains) over each dominant GEMM launch.
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r02
mkdir -p $O
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_WAIT_ANY"
P2="FETCH_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P3="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
run() {  # name, args...
  name=$1; shift
  i=1
  for P in "$P1" "$P2" "$P3"; do
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/pmc_$name -o p$i -- python3 $R/tools/gemm_one.py "$@" > $O/pmc_${name}_p$i.log 2>&1 || echo "pass $name p$i failed"
    i=$((i+1))
  done
  echo "done $name"
}
run f32_fwd --dtype f32 --shape fwd
run f32_wgrad --dtype f32 --shape wgrad
run bf16_fwd --dtype bf16 --shape fwd --out-bf16
run bf16_wgrad --dtype bf16 --shape wgrad
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o r02 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/trace.log 2>&1 || echo "trace failed"
ls -R $O | head -60
