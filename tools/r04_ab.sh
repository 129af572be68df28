#!/bin/bash
# round-4 A/B records (run on the GPU box from the repo root): non-temporal streams on / off in the headline step,
# forward fusion kernel with / without register prefetch.  Needs variants/libvqf_nt0.so (tools/build_variant.sh nt0
# "-DVQF_STREAM_NT=0" fusion.hip,attention.hip).
mkdir -p gpurun_out/r04
{
  echo "# headline step (python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-secondary), same box, alternating:"
  echo "# default = VQF_STREAM_NT=1 (non-temporal loads / stores on once-touched streams), nt0 = the same build with VQF_STREAM_NT=0"
  for v in default nt0 default nt0; do
    if [ $v = default ]; then unset VQF_LIB; else export VQF_LIB=variants/libvqf_$v.so; fi
    python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-secondary > /dev/null 2>&1
    echo "== $v"
    python tools/census_print.py | grep -E "ms_per_step|mfb_fuse|glimpse|att_logits|wgrad|fwd\)"
  done
  unset VQF_LIB
} > gpurun_out/r04/nt_ab.log 2>&1
{
  echo "# HBM-bound kernels alone at the headline shapes (tools/hbm_kernels_ab.py): default, then VQF_FUSE_COAL=1 (forward fusion kernel with register prefetch)"
  python tools/hbm_kernels_ab.py 2>&1 | grep -v amdgpu
  VQF_FUSE_COAL=1 python tools/hbm_kernels_ab.py 2>&1 | grep -E "library|fuse_fwd"
  python tools/hbm_kernels_ab.py 2>&1 | grep -E "library|fuse_fwd"
  VQF_FUSE_COAL=1 python tools/hbm_kernels_ab.py 2>&1 | grep -E "library|fuse_fwd"
} > gpurun_out/r04/hbm_ab.log 2>&1
tail -30 gpurun_out/r04/nt_ab.log
cat gpurun_out/r04/hbm_ab.log
