#!/bin/bash
# 3072- and 4096-bit keys: the ladder as shipped (--group 0) beside the 8- and 4-lane rungs pinned for every batch size
cd "$(dirname "$0")/../.."
out=gpurun_out/${TAG:-pinwide}; mkdir -p $out
for bits in ${BITS:-3072 4096}; do
  for grp in ${GROUPS_:-0 8 4}; do
    echo "== $bits bits, --group $grp" >> $out/sweep.txt
    timeout 400 python tools/bench_sweep.py --key-bits $bits --group $grp --min ${MIN:-13} --max ${MAX:-16} --ops ${OPS:-encrypt,decrypt,mul} --budget-ms 250 --table > $out/sweep_${bits}_g$grp.json 2>> $out/sweep.txt
  done
done
grep -v amdgpu.ids $out/sweep.txt
