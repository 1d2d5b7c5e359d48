#!/bin/bash
# Round 3, fourth GPU pass: the wave-pair kernels — GPU tests, the latency probe with and without them, the reference-shaped benchmark.
cd "$GRAFT_REPO_ROOT" || exit 1
T=${TAG:-r03d}
O=gpurun_out/$T; mkdir -p $O
timeout 300 python __graft_entry__.py smoke; echo "smoke rc=$?"
timeout 600 python -m pytest tests/test_gpu_ladder.py -m gpu -x -q -p no:cacheprovider > $O/pytest_ladder.txt 2>&1; echo "ladder rc=$?"; tail -5 $O/pytest_ladder.txt
timeout 600 python tools/bench_latency.py > $O/latency.json 2> $O/latency.txt; echo "latency rc=$?"; cat $O/latency.txt
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider --deselect tests/test_gpu_ladder.py > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.txt
timeout 900 python examples/benchmarks_batched.py --key-sizes 1024 2048 3072 4096 8192 > $O/benchmarks_batched.txt 2> $O/benchmarks_batched.err; echo "benchmarks rc=$?"; grep -E "key size|^encrypt|^decrypt" $O/benchmarks_batched.txt
timeout 300 python examples/federated_learning_batched.py > $O/federated_example_2048bit_gpu.log 2>&1; echo "federated rc=$?"; tail -4 $O/federated_example_2048bit_gpu.log
