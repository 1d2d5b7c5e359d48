#!/bin/bash
# decrypt sweep with the CRT tail on one wavefront per ciphertext at every batch size, beside the shipped threshold
cd "$(dirname "$0")/../.."
out=gpurun_out/${TAG:-wtail}; mkdir -p $out
for per_cu in ${PER_CUS:-16 1000000}; do
  echo "== PHE_HIP_WAVE_TAIL_PER_CU=$per_cu" >> $out/sweep.txt
  PHE_HIP_WAVE_TAIL_PER_CU=$per_cu timeout 300 python tools/bench_sweep.py --min ${MIN:-12} --max ${MAX:-18} --ops decrypt --budget-ms 250 --table \
     > $out/sweep_$per_cu.json 2>> $out/sweep.txt
done
cat $out/sweep.txt
