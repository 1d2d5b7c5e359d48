#!/bin/bash
# TAG=<name> bash tools/gpu_profile_round.sh (through gpurun): the round's measured record — smoke, bench (2048 with ops + the
# configs[3] leg / 3072 / 1024), the batch-size sweep, the reference's benchmark loop and the examples on the drop-in, API-level
# rates, the calibration microbenchmark, and the rocprofv3 passes: a kernel trace of the whole bench, kernel traces of an
# encrypt-only and a decrypt-only run (so that a kernel's average launch time belongs to ONE leg), and the PMC passes (own runs).
cd "$GRAFT_REPO_ROOT" || exit 1
T=${TAG:-r03}
O=gpurun_out/$T; mkdir -p $O; R=$PWD
timeout 200 python __graft_entry__.py smoke; echo "smoke rc=$?"
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench_1M.json 2> $O/bench_1M.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench_1M.json")); r=d["roofline"]
print(d["value"], d["decrypt"]["value"], r["frac"], r["canonical_frac"], r.get("mad_share_of_valu_instructions"), d["bit_exact"], d["cpu_baseline"]["value"])
for k,v in d["ops"].items(): print(" ", k, round(v["value"]), v.get("additions_per_s"), v["bit_exact_strided_sample_vs_gmp_oracle"])
c=d["config4"]; print(" config4", c and c["encrypt"], c and c["bit_exact_boundaries_and_sample_vs_gmp_oracle"])
PY
timeout 300 python bench.py --key-bits 3072 --batch 262144 --steps 1 --warmup 1 --no-cpu-baseline --no-config4 > $O/bench_3072.json 2>/dev/null; echo "3072 rc=$?"
timeout 300 python bench.py --key-bits 1024 --batch 1048576 --steps 1 --warmup 1 --no-cpu-baseline --no-config4 > $O/bench_1024.json 2>/dev/null; echo "1024 rc=$?"
timeout 400 python tools/bench_sweep.py --table > $O/batch_sweep.json 2> $O/batch_sweep.txt; echo "sweep rc=$?"; cat $O/batch_sweep.txt
timeout 900 python examples/benchmarks_batched.py > $O/benchmarks_batched.txt 2> $O/benchmarks_batched.err; echo "benchmarks rc=$?"; grep -v "^\[" $O/benchmarks_batched.txt | grep -E "key size|^encrypt|^decrypt"
timeout 300 python tools/bench_api.py 1048576 > $O/bench_api_1M.json 2>/dev/null; echo "api rc=$?"
timeout 300 python examples/federated_learning_batched.py > $O/federated_example_2048bit_gpu.log 2>&1; echo "federated rc=$?"; tail -3 $O/federated_example_2048bit_gpu.log
timeout 200 python-paillier_amd/lib/phe_microbench > $O/microbench.json 2> $O/microbench.err; echo "microbench rc=$?"
timeout 600 python tools/bench_latency.py > $O/latency.json 2> $O/latency.txt; echo "latency rc=$?"; cat $O/latency.txt
[ -x python-paillier_amd/lib/phe_latency_probe ] && timeout 120 python-paillier_amd/lib/phe_latency_probe > $O/latency_probe.json; echo "latency probe rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_kt -- python $R/bench.py --batch 262144 --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/prof_kt.log 2>&1; echo "kt rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_kt_enc -- python $R/bench.py --batch 262144 --steps 3 --warmup 1 --no-cpu-baseline --no-ops --no-config4 --only encrypt > $R/$O/prof_kt_enc.log 2>&1; echo "kt enc rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_kt_dec -- python $R/bench.py --batch 262144 --steps 3 --warmup 1 --no-cpu-baseline --no-ops --no-config4 --only decrypt > $R/$O/prof_kt_dec.log 2>&1; echo "kt dec rc=$?"
timeout 200 rocprofv3 --kernel-trace --stats -d $R/$O/prof_kt_lat -- python $R/tools/lat_one_probe.py 2048 > $R/$O/prof_kt_lat.log 2>&1; echo "kt latency rc=$?"
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS -d $R/$O/prof_pmc1 -- python $R/bench.py --batch 131072 --steps 1 --warmup 0 --no-cpu-baseline --no-config4 > $R/$O/prof_pmc1.log 2>&1; echo "pmc1 rc=$?"
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $R/$O/prof_pmc2 -- python $R/bench.py --batch 131072 --steps 1 --warmup 0 --no-cpu-baseline --no-config4 > $R/$O/prof_pmc2.log 2>&1; echo "pmc2 rc=$?"
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $R/$O/prof_pmc3 -- python $R/bench.py --batch 131072 --steps 1 --warmup 0 --no-cpu-baseline --no-config4 > $R/$O/prof_pmc3.log 2>&1; echo "pmc3 rc=$?"
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES -d $R/$O/prof_mb -- $R/python-paillier_amd/lib/phe_microbench > $R/$O/prof_mb.log 2>&1; echo "pmc microbench rc=$?"
cd $R
python tools/rocprof_summarize.py $O/prof_kt > $O/rocprofv3_kernel_trace_stats.txt 2>&1
python tools/rocprof_summarize.py $O/prof_kt_enc > $O/rocprofv3_kernel_trace_stats_encrypt_only.txt 2>&1
python tools/rocprof_summarize.py $O/prof_kt_dec > $O/rocprofv3_kernel_trace_stats_decrypt_only.txt 2>&1
python tools/rocprof_summarize.py $O/prof_kt_lat > $O/rocprofv3_kernel_trace_stats_one_decrypt.txt 2>&1
python tools/rocprof_summarize.py $O/prof_pmc1 $O/prof_pmc2 $O/prof_pmc3 > $O/rocprofv3_pmc.txt 2>&1
PHE_SUMMARIZE_ALL=1 python tools/rocprof_summarize.py $O/prof_mb > $O/microbench_rocprofv3_pmc.txt 2>&1
rm -rf $O/prof_kt $O/prof_kt_enc $O/prof_kt_dec $O/prof_kt_lat $O/prof_pmc1 $O/prof_pmc2 $O/prof_pmc3 $O/prof_mb
head -30 $O/rocprofv3_kernel_trace_stats.txt; cat $O/rocprofv3_kernel_trace_stats_decrypt_only.txt | head -30; grep -E "SQ_INSTS_VALU|FETCH_SIZE|WRITE_SIZE" $O/rocprofv3_pmc.txt | head -30
