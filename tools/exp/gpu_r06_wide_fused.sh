#!/bin/bash
# 3072-bit keys, same box: the round-5 library (libphe_hip_base.so: every pair product of the L = 27 rungs as single sweeps) against the
# tree's (squarings and conversions fused: split_core.h kMaxFusedPass2L), rungs pinned, 2^18 rows, interleaved twice.
cd "$(dirname "$0")/../.."
out=gpurun_out/${TAG:-r06b}; mkdir -p $out
export PHE_HIP_NO_MEASURED_LADDER=1
for rep in 1 2; do
for lib in libphe_hip_base.so libphe_hip.so; do
  for grp in 2 4; do
    echo "== $lib --group $grp (rep $rep)" >> $out/wide_fused.txt
    PHE_HIP_LIB=$PWD/python-paillier_amd/lib/$lib timeout 300 python tools/bench_sweep.py --key-bits 3072 --group $grp --min ${MIN:-18} --max ${MAX:-18} --ops encrypt,decrypt,mul --budget-ms 1500 --table > $out/wf_${lib%.so}_g${grp}_$rep.json 2>> $out/wide_fused.txt
  done
done
done
grep -v amdgpu $out/wide_fused.txt
