#!/bin/bash
# bash tools/gpu_ab_window.sh (through gpurun): A/B of the sliding-window width of the batch-uniform exponents (PHE_HIP_WINDOW)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for KB in 2048 3072; do
  for W in 5 6 7 4; do
    B=524288; [ $KB = 3072 ] && B=131072
    PHE_HIP_WINDOW=$W timeout 300 python bench.py --key-bits $KB --batch $B --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('key', $KB, 'window', $W, 'enc/s %.0f' % d['value'], 'dec/s %.0f' % d['decrypt']['value'], d['bit_exact'])" | tee -a gpurun_out/ab_window.txt
  done
done
