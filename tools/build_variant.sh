#!/bin/bash
# usage: tools/build_variant.sh NAME "-DVQF_GEMM_BK=16 ..." [file.hip]   -> variants/libvqf_NAME.so
# Recompiles ONE translation unit (default gemm_f32.hip) with extra flags and links it with the
# objects of the regular build (run `make -C vqa-attention-networks_amd/csrc` first).
# Load it with VQF_LIB=variants/libvqf_NAME.so (host/lib.py) or tools/gemm_ab.py.
set -e
cd "$(dirname "$0")/../vqa-attention-networks_amd/csrc"
mkdir -p ../../variants
src=${3:-gemm_f32.hip}
obj=/tmp/vqf_${1}_${src%.hip}.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $2 -c $src -o $obj
others=$(sed -n 's/^SRCS := //p' Makefile | tr ' ' '\n' | grep -v "^$src$" | sed 's/\.hip$/.o/' | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $obj $others -o ../../variants/libvqf_$1.so
echo built variants/libvqf_$1.so
