#!/bin/bash
# 3072-bit key: the throughput geometries the ladder ends on (G=4 x 27 limbs, G=2 x 27 for the CRT halves: too wide for the fused
# sweeps) against the next rung (fused sweeps) pinned for every batch size
cd "$(dirname "$0")/../.."
out=gpurun_out/${TAG:-pin3072}; mkdir -p $out
for grp in 0 8 4; do
  echo "== 3072 bits, --group $grp" >> $out/sweep.txt
  timeout 400 python tools/bench_sweep.py --key-bits 3072 --group $grp --min ${MIN:-13} --max ${MAX:-17} --ops ${OPS:-encrypt,decrypt,mul} --budget-ms 300 --table > $out/sweep_g$grp.json 2>> $out/sweep.txt
done
cat $out/sweep.txt
