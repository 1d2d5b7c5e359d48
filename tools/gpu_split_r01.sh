#!/bin/bash
# round-1 check of the split-modulus engine on the GPU: parity suite, then bench with both engines
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/split_tests.log
cat gpurun_out/split_tests.log
timeout 600 python bench.py --steps 2 --warmup 1 > gpurun_out/bench_split.json 2> gpurun_out/bench_split.err
tail -c 3000 gpurun_out/bench_split.json
PHE_HIP_ENGINE=full timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
tail -c 1500 gpurun_out/bench_full.json
