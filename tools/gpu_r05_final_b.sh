#!/bin/bash
# TAG=<name> bash tools/gpu_r05_final_b.sh (through gpurun): second half of the round-5 record — kernel traces of the driver's line at the full
# batch and of a decrypt-only run, the batch sweeps at three key widths, the scalar-loop shape of configs[4]'s file, the reference's
# benchmark loop, scalar latencies, the API-level rates.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
R=$PWD; O=gpurun_out/${TAG:-r05l}; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_kt -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/prof_kt.log 2>&1; echo "kt rc=$?")
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_kt_dec -- python $R/bench.py --batch 262144 --steps 3 --warmup 1 --no-cpu-baseline --no-ops --no-config4 --only decrypt > $R/$O/prof_kt_dec.log 2>&1; echo "kt dec rc=$?")
python tools/rocprof_summarize.py $O/prof_kt > $O/rocprofv3_kernel_trace_stats_1M.txt 2>&1; python tools/rocprof_summarize.py $O/prof_kt_dec > $O/rocprofv3_kernel_trace_stats_decrypt_only.txt 2>&1
grep -h '^{' $O/prof_kt.log | tail -1 > $O/bench_1M_under_tracer.json
rm -rf $O/prof_kt $O/prof_kt_dec
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_1M_same_box_untraced.json 2>/dev/null; echo "untraced rc=$?"
timeout 300 python tools/bench_sweep.py --table > $O/batch_sweep.json 2> $O/batch_sweep.txt; echo "sweep rc=$?"; grep -v amdgpu $O/batch_sweep.txt
timeout 200 python tools/bench_sweep.py --key-bits 1024 --min 10 --max 20 --ops encrypt,decrypt,add,mul --budget-ms 200 --table > $O/batch_sweep_1024.json 2> $O/batch_sweep_1024.txt; grep -v amdgpu $O/batch_sweep_1024.txt
timeout 250 python tools/bench_sweep.py --key-bits 3072 --min 10 --max 18 --ops encrypt,decrypt,add,mul --budget-ms 250 --table > $O/batch_sweep_3072.json 2> $O/batch_sweep_3072.txt; grep -v amdgpu $O/batch_sweep_3072.txt
timeout 300 python tools/federated_scalar_shape.py > $O/federated_scalar_shape.json 2> $O/federated_scalar_shape.err; echo "scalar shape rc=$?"; head -c 1500 $O/federated_scalar_shape.json; echo
timeout 400 python tools/bench_latency.py > $O/latency.json 2> $O/latency.txt; echo "latency rc=$?"; cat $O/latency.txt
timeout 600 python examples/benchmarks_batched.py > $O/benchmarks_batched.txt 2> $O/benchmarks_batched.err; echo "benchmarks rc=$?"; grep -v "^\[" $O/benchmarks_batched.txt | grep -E "key size|^encrypt|^decrypt|^add enc"
timeout 200 python tools/bench_api.py > $O/bench_api.json 2>/dev/null; echo "api rc=$?"
