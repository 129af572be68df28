#!/bin/bash
# usage: tools/build_variant.sh NAME "-DVQF_GEMM_BK=16 ..." [file.hip[,file2.hip...]]   -> variants/libvqf_NAME.so
# Recompiles the named translation units (default gemm_f32.hip) with extra flags and links them with the
# objects of the regular build (run `make -C vqa-attention-networks_amd/csrc` first).
# Load it with VQF_LIB=variants/libvqf_NAME.so (host/lib.py) or tools/gemm_ab.py.
set -e
cd "$(dirname "$0")/../vqa-attention-networks_amd/csrc"
mkdir -p ../../variants
srcs=$(echo "${3:-gemm_f32.hip}" | tr ',' ' ')
objs=""
others=$(sed -n 's/^SRCS := //p' Makefile | tr ' ' '\n' | sed 's/\.hip$/.o/' | tr '\n' ' ')
for src in $srcs; do
  obj=/tmp/vqf_${1}_${src%.hip}.o
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $2 -c $src -o $obj
  objs="$objs $obj"
  others=$(echo "$others" | tr ' ' '\n' | grep -v "^${src%.hip}.o$" | tr '\n' ' ')
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $others -o ../../variants/libvqf_$1.so
echo built variants/libvqf_$1.so
