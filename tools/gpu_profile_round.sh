#!/bin/bash
# TAG=<name> bash tools/gpu_profile_round.sh (through gpurun): smoke, bench (2048 with ops / 3072 / 1024), the reference's
# benchmark loop and the examples on the drop-in, API-level rates, rocprofv3 kernel trace and PMC passes (own runs).
cd "$GRAFT_REPO_ROOT" || exit 1
T=${TAG:-r02}
O=gpurun_out/$T; mkdir -p $O; R=$PWD
timeout 200 python __graft_entry__.py smoke; echo "smoke rc=$?"
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench_1M.json 2> $O/bench_1M.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench_1M.json")); r=d["roofline"]
print(d["value"], d["decrypt"]["value"], r["frac"], r["canonical_frac"], r["mad_share_of_valu_instructions"], d["bit_exact"], d["cpu_baseline"]["value"])
for k,v in d["ops"].items(): print(" ", k, round(v["value"]), v.get("additions_per_s"), v["bit_exact_strided_sample_vs_gmp_oracle"])
PY
timeout 300 python bench.py --key-bits 3072 --batch 262144 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_3072.json 2>/dev/null; echo "3072 rc=$?"
timeout 300 python bench.py --key-bits 1024 --batch 1048576 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_1024.json 2>/dev/null; echo "1024 rc=$?"
timeout 600 python examples/benchmarks_batched.py > $O/benchmarks_batched.txt 2> $O/benchmarks_batched.err; echo "benchmarks rc=$?"; tail -9 $O/benchmarks_batched.txt | head -8
timeout 300 python tools/bench_api.py 1048576 > $O/bench_api_1M.json 2>/dev/null; echo "api rc=$?"
timeout 300 python examples/federated_learning_batched.py > $O/federated_example_2048bit_gpu.log 2>&1; echo "federated rc=$?"; tail -3 $O/federated_example_2048bit_gpu.log
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_kt -- python $R/bench.py --batch 262144 --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/prof_kt.log 2>&1; echo "kt rc=$?"
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS -d $R/$O/prof_pmc1 -- python $R/bench.py --batch 131072 --steps 1 --warmup 0 --no-cpu-baseline > $R/$O/prof_pmc1.log 2>&1; echo "pmc1 rc=$?"
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $R/$O/prof_pmc2 -- python $R/bench.py --batch 131072 --steps 1 --warmup 0 --no-cpu-baseline > $R/$O/prof_pmc2.log 2>&1; echo "pmc2 rc=$?"
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $R/$O/prof_pmc3 -- python $R/bench.py --batch 131072 --steps 1 --warmup 0 --no-cpu-baseline > $R/$O/prof_pmc3.log 2>&1; echo "pmc3 rc=$?"
cd $R
python tools/rocprof_summarize.py $O/prof_kt > $O/rocprofv3_kernel_trace_stats.txt 2>&1
python tools/rocprof_summarize.py $O/prof_pmc1 $O/prof_pmc2 $O/prof_pmc3 > $O/rocprofv3_pmc.txt 2>&1
rm -rf $O/prof_kt $O/prof_pmc1 $O/prof_pmc2 $O/prof_pmc3
head -24 $O/rocprofv3_kernel_trace_stats.txt; grep -E "SQ_INSTS_VALU|FETCH_SIZE|WRITE_SIZE" $O/rocprofv3_pmc.txt | head -30
