#!/bin/bash
# TAG=<name> bash tools/gpu_profile_round_lean.sh (through gpurun): the short form of gpu_profile_round.sh for the end of a round
# — smoke, the ladder tests, the driver's bench line, the 3072-bit bench, the batch sweeps, the reference's benchmark loop, scalar
# latencies, and the kernel traces of the whole bench and of a decrypt-only run.  No PMC passes, no microbenchmark.
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
O=gpurun_out/${TAG:-lean}; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python __graft_entry__.py smoke; echo "smoke rc=$?"
timeout 300 python -m pytest tests/test_gpu_ladder.py -m gpu -x -q > $O/pytest_ladder.txt 2>&1; grep -E "passed|failed" $O/pytest_ladder.txt
timeout 600 python bench.py > $O/bench_1M.json 2> $O/bench_1M.err; echo "bench rc=$?"
timeout 300 python bench.py --key-bits 3072 --batch 262144 --steps 1 --warmup 1 --no-cpu-baseline --no-config4 > $O/bench_3072.json 2>/dev/null; echo "3072 rc=$?"
timeout 300 python tools/bench_sweep.py --table > $O/batch_sweep.json 2> $O/batch_sweep.txt; echo "sweep rc=$?"; grep -v amdgpu $O/batch_sweep.txt
timeout 200 python tools/bench_sweep.py --key-bits 1024 --min 10 --max 17 --ops encrypt,decrypt --budget-ms 200 --table > $O/batch_sweep_1024.json 2> $O/batch_sweep_1024.txt; grep -v amdgpu $O/batch_sweep_1024.txt
timeout 200 python tools/bench_sweep.py --key-bits 3072 --min 12 --max 17 --ops encrypt,decrypt,mul --budget-ms 250 --table > $O/batch_sweep_3072.json 2> $O/batch_sweep_3072.txt; grep -v amdgpu $O/batch_sweep_3072.txt
timeout 200 python tools/scalar_op_breakdown.py 2048 2>&1 | head -9 > $O/scalar_breakdown_2048.txt; cat $O/scalar_breakdown_2048.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_kt_dec -- python $R/bench.py --batch 262144 --steps 3 --warmup 1 --no-cpu-baseline --no-ops --no-config4 --only decrypt > $R/$O/prof_kt_dec.log 2>&1; echo "kt dec rc=$?")
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_kt -- python $R/bench.py --batch 262144 --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/prof_kt.log 2>&1; echo "kt rc=$?")
python tools/rocprof_summarize.py $O/prof_kt > $O/rocprofv3_kernel_trace_stats.txt 2>&1; python tools/rocprof_summarize.py $O/prof_kt_dec > $O/rocprofv3_kernel_trace_stats_decrypt_only.txt 2>&1
timeout 400 python tools/bench_latency.py > $O/latency.json 2> $O/latency.txt; echo "latency rc=$?"; cat $O/latency.txt
timeout 600 python examples/benchmarks_batched.py > $O/benchmarks_batched.txt 2> $O/benchmarks_batched.err; echo "benchmarks rc=$?"; grep -v "^\[" $O/benchmarks_batched.txt | grep -E "key size|^encrypt|^decrypt|^add enc"
