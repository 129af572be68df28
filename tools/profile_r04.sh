#!/bin/bash
# rocprofv3 evidence of round 4 (run on the GPU box from the repo root).  PMC passes are separate runs with
# --kernel-trace only (never combined with other trace domains); the program follows `--` directly.
#   part "hie":   the same over the streaming passes of HieCoAtten's ladder (tools/hie_kernels_one.py)
#   part "hbm":   three PMC passes over the HBM-bound kernels at the headline shapes (tools/hbm_kernels_one.py)
#   part "trace": kernel trace + stats of the bench step (headline + secondary configs 3 and 4 in the same process)
#   part "gemm":  three PMC passes over the dominant GEMM launches (tools/gemm_one.py)
#   part "mid":   the same over the row-split mid-size products (HieCoAtten img_emb forward / weight gradient, co_att_conv1 forward)
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r04
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
    hbm)   pmc hbm $R/tools/hbm_kernels_one.py ;;
    hie)   pmc hie $R/tools/hie_kernels_one.py ;;
    gemm)  pmc f32_fwd $R/tools/gemm_one.py --dtype f32 --shape fwd
           pmc f32_wgrad $R/tools/gemm_one.py --dtype f32 --shape wgrad
           pmc bf16_fwd $R/tools/gemm_one.py --dtype bf16 --shape fwd --out-bf16
           pmc bf16_wgrad $R/tools/gemm_one.py --dtype bf16 --shape wgrad ;;
    mid)   pmc hie_fwd $R/tools/gemm_one.py --dtype f32 --shape hie_fwd
           pmc hie_wgrad $R/tools/gemm_one.py --dtype f32 --shape hie_wgrad
           pmc coatt_fwd $R/tools/gemm_one.py --dtype f32 --shape coatt_fwd ;;
    trace) timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o r04 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --secondary-steps 5 --secondary-warmup 2 > $O/trace.log 2>&1 || echo "trace failed" ;;
  esac
done
ls -R $O | head -80
