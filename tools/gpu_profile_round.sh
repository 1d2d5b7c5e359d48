#!/bin/bash
# TAG=<name> bash tools/gpu_profile_round.sh (through gpurun): parity suite, smoke, bench (2048 / 3072 / ops), rocprofv3 kernel trace and PMC passes for the split-modulus engine
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; R=$PWD
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_${TAG:-r01}.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_${TAG:-r01}.log
timeout 200 python __graft_entry__.py smoke; echo "smoke rc=$?"
timeout 600 python bench.py > gpurun_out/bench_1M_${TAG:-r01}.json 2> gpurun_out/bench_1M_${TAG:-r01}.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/bench_1M_${TAG:-r01}.json')); print(d['value'], d['decrypt']['value'], d['roofline']['frac'], d['roofline']['executed'], d['bit_exact'], d['cpu_baseline']['value'], d['cpu_baseline']['decrypts_per_s'], d['config']['geometry'])"
timeout 300 python bench.py --key-bits 3072 --batch 262144 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_3072_${TAG:-r01}.json 2>/dev/null; echo "3072 rc=$?"
timeout 300 python bench.py --key-bits 1024 --batch 1048576 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_1024_${TAG:-r01}.json 2>/dev/null; echo "1024 rc=$?"
timeout 300 python tools/bench_ops.py --batch 1048576 > gpurun_out/bench_ops_${TAG:-r01}.json 2>/dev/null; echo "ops rc=$?"
timeout 300 python tools/bench_wire.py > gpurun_out/bench_wire_${TAG:-r01}.json 2>/dev/null; echo "wire rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -- python $R/bench.py --batch 262144 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_kt.log 2>&1; echo "kt rc=$?"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS -d $R/gpurun_out/prof_pmc1 -- python $R/bench.py --batch 131072 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof_pmc1.log 2>&1; echo "pmc1 rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_pmc2 -- python $R/bench.py --batch 131072 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof_pmc2.log 2>&1; echo "pmc2 rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_pmc3 -- python $R/bench.py --batch 131072 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof_pmc3.log 2>&1; echo "pmc3 rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt_ops -- python $R/tools/bench_ops.py --batch 262144 --reps 1 > $R/gpurun_out/prof_kt_ops.log 2>&1; echo "kt ops rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt_wire -- python $R/tools/bench_wire.py --batch 65536 > $R/gpurun_out/prof_kt_wire.log 2>&1; echo "kt wire rc=$?"
cd $R
python tools/rocprof_summarize.py gpurun_out/prof_kt_ops gpurun_out/prof_kt_wire > gpurun_out/${TAG:-r01}_rocprofv3_kernel_trace_stats_ops_wire.txt 2>&1
rm -rf gpurun_out/prof_kt_ops gpurun_out/prof_kt_wire
python tools/rocprof_summarize.py gpurun_out/prof_kt > gpurun_out/${TAG:-r01}_rocprofv3_kernel_trace_stats.txt 2>&1
python tools/rocprof_summarize.py gpurun_out/prof_pmc1 gpurun_out/prof_pmc2 gpurun_out/prof_pmc3 > gpurun_out/${TAG:-r01}_rocprofv3_pmc.txt 2>&1
rm -rf gpurun_out/prof_kt gpurun_out/prof_pmc1 gpurun_out/prof_pmc2 gpurun_out/prof_pmc3
head -30 gpurun_out/${TAG:-r01}_rocprofv3_kernel_trace_stats.txt; cat gpurun_out/${TAG:-r01}_rocprofv3_pmc.txt | head -60
