#!/bin/bash
# Measurement-only builds of the library with other COMPILER flags for the translation units of the 2048-bit encrypt / decrypt
# geometries (kernels_s4b: 4 x 18, kernels_s2b: 2 x 18); everything else is linked from build/obj.
# usage: bash tools/exp/build_flag_variants.sh name1 "flags1" name2 "flags2" ...  ->  python-paillier_amd/lib/libphe_hip_<name>.so
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CSRC=$ROOT/python-paillier_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=1000000 -fPIC"
names=()
while [ $# -ge 2 ]; do
  V=$1; X=$2; shift 2; names+=($V)
  D=$ROOT/build/obj_$V; mkdir -p $D
  for U in kernels_s4b kernels_s2b; do
    /opt/rocm/bin/hipcc $FLAGS $X -c -o $D/$U.o $CSRC/$U.hip &
  done
done
wait
for V in "${names[@]}"; do
  D=$ROOT/build/obj_$V
  OBJS=$(ls $ROOT/build/obj/*.o | grep -v -e kernels_s4b.o -e kernels_s2b.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/python-paillier_amd/lib/libphe_hip_$V.so $OBJS $D/kernels_s4b.o $D/kernels_s2b.o
  echo built libphe_hip_$V.so
done
