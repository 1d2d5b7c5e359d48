#!/bin/bash
# same-box A/B of several builds of the library (LIBS="a.so b.so ..." under python-paillier_amd/lib), interleaved REPS times:
# encrypt / decrypt (OPS) at 2^MIN..2^MAX rows of a BITS-bit key through tools/bench_sweep.py (every point checked against libgmp)
cd "$(dirname "$0")/../.."
out=gpurun_out/${TAG:-ab}; mkdir -p $out
for rep in $(seq 1 ${REPS:-2}); do
  for lib in $LIBS; do
    echo "== $lib (rep $rep)" >> $out/ab.txt
    PHE_HIP_LIB=$PWD/python-paillier_amd/lib/$lib timeout 400 python tools/bench_sweep.py --key-bits ${BITS:-2048} --min ${MIN:-18} --max ${MAX:-18} --ops ${OPS:-encrypt,decrypt} --budget-ms ${BUDGET:-3000} --table > $out/ab_${lib%.so}_$rep.json 2>> $out/ab.txt
  done
done
grep -v amdgpu $out/ab.txt
