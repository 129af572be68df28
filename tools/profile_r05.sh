#!/bin/bash
# rocprofv3 evidence of round 5 (run on the GPU box from the repo root).  PMC passes are separate runs with --kernel-trace only
# (never combined with other trace domains); the program follows `--` directly.
#   part "order": the split-K work-item order A/B (library option gemm_splitk_order, read from the environment at load): three
#                 PMC passes over the image projection's weight gradient, fp32 and bf16, order 0 and 1
#   part "gemm":  the dominant GEMM launches (forward + weight gradient, fp32 + bf16) with the default options
#   part "hie":   the streaming passes of HieCoAtten's ladder
#   part "hbm":   the HBM-bound kernels at the headline shapes
#   part "trace": kernel trace + stats of the default bench command (headline + secondary configs)
#   part "c4":    the per-sample-tile GEMM launches of config 4
#   part "n80":   the one-round 128x80-tile kernel on the 512 x 5000 x 2048 projection
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r05
mkdir -p $O
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_WAIT_ANY"
P2="FETCH_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P3="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
pmc() {  # name, program args...
  name=$1; shift
  i=1
  for P in "$P1" "$P2" "$P3"; do
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/pmc_$name -o p$i -- python3 "$@" > $O/pmc_${name}_p$i.log 2>&1 || echo "pass $name p$i failed"
    i=$((i+1))
  done
  echo "done $name"
}
for part in "$@"; do
  case $part in
    order) for o in 0 1; do
             export VQF_GEMM_SPLITK_ORDER=$o
             pmc f32_wgrad_o$o $R/tools/gemm_one.py --dtype f32 --shape wgrad
             pmc bf16_wgrad_o$o $R/tools/gemm_one.py --dtype bf16 --shape wgrad
           done
           unset VQF_GEMM_SPLITK_ORDER ;;
    gemm)  pmc f32_fwd $R/tools/gemm_one.py --dtype f32 --shape fwd
           pmc f32_wgrad $R/tools/gemm_one.py --dtype f32 --shape wgrad
           pmc bf16_fwd $R/tools/gemm_one.py --dtype bf16 --shape fwd --out-bf16
           pmc bf16_wgrad $R/tools/gemm_one.py --dtype bf16 --shape wgrad ;;
    hie)   pmc hie $R/tools/hie_kernels_one.py ;;
    hbm)   pmc hbm $R/tools/hbm_kernels_one.py ;;
    n80)   pmc m512 $R/tools/gemm_one.py --dtype f32 --shape m512 --reps 10 ;;
    c4)    pmc hie_fwd $R/tools/gemm_one.py --dtype f32 --shape hie_fwd
           pmc hie_dgrad $R/tools/gemm_one.py --dtype f32 --shape hie_dgrad ;;
    trace) timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o r05 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --secondary-steps 5 --secondary-warmup 2 > $O/trace.log 2>&1 || echo "trace failed" ;;
  esac
done
ls $O | head -60
