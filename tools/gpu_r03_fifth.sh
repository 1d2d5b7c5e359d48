#!/bin/bash
# Round 3, fifth GPU pass: the wave-pair kernels on the scaled modulus (quotient digit off the critical chain) — ladder tests, latency probe.
cd "$GRAFT_REPO_ROOT" || exit 1
T=${TAG:-r03e}
O=gpurun_out/$T; mkdir -p $O
timeout 300 python __graft_entry__.py smoke; echo "smoke rc=$?"
timeout 600 python -m pytest tests/test_gpu_ladder.py -m gpu -x -q -p no:cacheprovider > $O/pytest_ladder.txt 2>&1; echo "ladder rc=$?"; tail -5 $O/pytest_ladder.txt
timeout 600 python tools/bench_latency.py > $O/latency.json 2> $O/latency.txt; echo "latency rc=$?"; cat $O/latency.txt
