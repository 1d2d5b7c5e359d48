#!/bin/bash
# same-box A/B of two builds of the library (PHE_HIP_LIB): batch sweep with each, alternating
cd "$(dirname "$0")/../.."
out=gpurun_out/${TAG:-ab}; mkdir -p $out
A=${LIB_A:-python-paillier_amd/lib/libphe_hip_base.so}; B=${LIB_B:-python-paillier_amd/lib/libphe_hip.so}
for lib in $A $B $A $B; do
  echo "== $lib" >> $out/sweep.txt
  PHE_HIP_LIB=$PWD/$lib timeout 400 python tools/bench_sweep.py --key-bits ${BITS:-2048} --min ${MIN:-10} --max ${MAX:-16} --ops ${OPS:-encrypt,decrypt,mul} --budget-ms ${BUDGET:-250} --table > $out/sweep_$(basename $lib .so)_$RANDOM.json 2>> $out/sweep.txt
done
grep -v amdgpu.ids $out/sweep.txt
