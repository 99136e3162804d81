#!/bin/bash
# Builds the ablation variants of csrc/gemm.hip (CPCSV_PROBE bits 16 / 32 / 64, see gemm.hip) into tools/probe/_build/ - on the CPU
# container (hipcc cross-compiles); the .so files travel to the GPU box with the snapshot. Then: python tools/nt_ablate.py
set -e
cd "$(dirname "$0")/.."
PKG=cpcstoryvisualization-pytorch_amd
OUT=tools/probe/_build
mkdir -p $OUT
make -C $PKG/csrc >/dev/null
build() {
  v=$1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DCPCSV_PROBE=$v -Iinclude -c $PKG/csrc/gemm.hip -o $OUT/gemm_p$v.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OUT/gemm_p$v.o $PKG/csrc/norm.o $PKG/csrc/elementwise.o $PKG/csrc/small.o \
      $PKG/csrc/thin.o $PKG/csrc/abi.o $PKG/csrc/head.o -o $OUT/libcpcsv_p$v.so
  rm -f $OUT/gemm_p$v.o
  echo built $v
}
for v in ${VARIANTS:-16 32 64 48 80 96 112}; do build $v & if (( $(jobs -r | wc -l) >= 4 )); then wait -n; fi; done
wait
