#!/bin/bash
# Measurement-only builds of the library with a variant of the pair sweeps (csrc/split_core.h PHE_VARIANT_*): only the
# translation units of the 2048-bit encrypt / decrypt geometries are recompiled, the rest is linked from build/obj.
# usage: bash tools/exp/build_variants.sh QMAD UNITINV   ->  python-paillier_amd/lib/libphe_hip_<variant>.so
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CSRC=$ROOT/python-paillier_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=1000000 -fPIC"
for V in "$@"; do
  D=$ROOT/build/obj_$V; mkdir -p $D
  for U in kernels_s4b kernels_s2b; do
    /opt/rocm/bin/hipcc $FLAGS -DPHE_VARIANT_$V -c -o $D/$U.o $CSRC/$U.hip &
  done
  wait
  OBJS=$(ls $ROOT/build/obj/*.o | grep -v -e kernels_s4b.o -e kernels_s2b.o)
  low=$(echo $V | tr A-Z a-z)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/python-paillier_amd/lib/libphe_hip_$low.so $OBJS $D/kernels_s4b.o $D/kernels_s2b.o
  echo built libphe_hip_$low.so
done
