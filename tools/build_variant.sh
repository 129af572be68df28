#!/bin/bash
# usage: tools/build_variant.sh NAME "-DVQF_GEMM_BK=16 ..."   -> gpurun_out/variants/libvqf_NAME.so
set -e
cd "$(dirname "$0")/../vqa-attention-networks_amd/csrc"
mkdir -p ../../variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $2 -c gemm_f32.hip -o /tmp/gemm_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/gemm_$1.o gemm_bf16.o attention.o fusion.o reduce.o elementwise.o lstm.o prof.o -o ../../variants/libvqf_$1.so
echo built variants/libvqf_$1.so
