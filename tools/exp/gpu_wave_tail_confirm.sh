#!/bin/bash
# after the default changed: GPU suite, decrypt sweeps per key size with the old rule (16 per CU) beside the new default, bench line
cd "$(dirname "$0")/../.."
out=gpurun_out/${TAG:-wtail2}; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1; tail -3 $out/pytest_gpu.txt
for bits in 1024 3072 4096; do
  for per_cu in 16 default; do
    echo "== $bits bits, PHE_HIP_WAVE_TAIL_PER_CU=$per_cu" >> $out/sweep_keys.txt
    if [ $per_cu = default ]; then unset PHE_HIP_WAVE_TAIL_PER_CU; else export PHE_HIP_WAVE_TAIL_PER_CU=$per_cu; fi
    timeout 300 python tools/bench_sweep.py --key-bits $bits --min 12 --max 17 --ops decrypt --budget-ms 200 --table > $out/sweep_${bits}_$per_cu.json 2>> $out/sweep_keys.txt
  done
done
unset PHE_HIP_WAVE_TAIL_PER_CU
cat $out/sweep_keys.txt
timeout 600 python bench.py > $out/bench_1M.json 2> $out/bench_1M.err; tail -c 600 $out/bench_1M.json
