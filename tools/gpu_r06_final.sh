#!/bin/bash
# TAG=<name> bash tools/gpu_r06_final.sh (through gpurun): the measurements of record of round 6 on the final device sources —
# the whole GPU suite, the driver's bench line (+ 1024 / 3072 bits), EVERY row of the 2^20-row headline batch against libgmp, the kernel
# trace of the bench command, the PMC passes behind roofline.traffic, batch sweeps, scalar latencies, API-level rates, configs[4].
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
O=gpurun_out/${TAG:-r06f}; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python __graft_entry__.py smoke; echo "smoke rc=$?"
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.txt | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_1M.json 2> $O/bench_1M.err; echo "bench rc=$?"
timeout 400 python bench.py --key-bits 1024 --steps 5 --warmup 2 --no-config4 > $O/bench_1024.json 2>/dev/null; echo "1024 rc=$?"
timeout 400 python bench.py --key-bits 3072 --batch 262144 --steps 2 --warmup 1 --no-config4 > $O/bench_3072.json 2>/dev/null; echo "3072 rc=$?"
# every row of the headline batch against libgmp (not a bench line: one step, the oracle takes ~6 minutes on 16 cores)
timeout 900 python bench.py --steps 1 --warmup 0 --oracle-sample 1048576 --no-ops --no-config4 --no-cpu-baseline > $O/every_row_vs_libgmp_2048_1M.json 2>/dev/null; echo "every-row rc=$?"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_kt -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/bench_1M_under_tracer.json 2> $R/$O/prof_kt.log; echo "kt rc=$?")
python tools/rocprof_summarize.py $O/prof_kt > $O/rocprofv3_kernel_trace_stats_1M.txt 2>&1; rm -rf $O/prof_kt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_kt_dec -- python $R/bench.py --batch 262144 --steps 3 --warmup 1 --no-cpu-baseline --no-ops --no-config4 --only decrypt > $R/$O/prof_kt_dec.log 2>&1; echo "kt dec rc=$?")
python tools/rocprof_summarize.py $O/prof_kt_dec > $O/rocprofv3_kernel_trace_stats_decrypt_only.txt 2>&1; rm -rf $O/prof_kt_dec
TAG=${TAG:-r06f} GRAFT_REPO_ROOT=$R bash tools/gpu_pmc_traffic.sh > $O/pmc.log 2>&1; echo "pmc rc=$?"; tail -5 $O/pmc.log
timeout 300 python tools/bench_sweep.py --table > $O/batch_sweep.json 2> $O/batch_sweep.txt; echo "sweep rc=$?"; grep -v amdgpu $O/batch_sweep.txt
timeout 400 python tools/bench_latency.py > $O/latency.json 2> $O/latency.txt; echo "latency rc=$?"; cat $O/latency.txt
timeout 300 python tools/bench_api.py 1048576 > $O/bench_api_1M.json 2>/dev/null; echo "api rc=$?"
timeout 400 python -m pytest tests/test_federated_example.py -m gpu -q -s -k scalar > $O/federated_scalar_api_gpu.txt 2>&1; echo "federated rc=$?"
timeout 300 python tools/calibrate_ladder.py --check python-paillier_amd/phe/ladder_gfx950.txt > $O/ladder_check.json 2>/dev/null; echo "ladder check rc=$?"; tail -c 600 $O/ladder_check.json
