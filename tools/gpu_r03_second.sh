#!/bin/bash
# Round 3, second GPU pass (through gpurun): GPU tests, the sweep on the cost-model ladder (+ the whole-wave rung pinned), the
# reference-shaped benchmark (scalar latencies), the calibration microbenchmark with the MFMA-beside-VALU experiment.
cd "$GRAFT_REPO_ROOT" || exit 1
T=${TAG:-r03b}
O=gpurun_out/$T; mkdir -p $O; R=$PWD
timeout 300 python __graft_entry__.py smoke; echo "smoke rc=$?"
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider -p no:faulthandler > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_gpu.txt
timeout 400 python tools/bench_sweep.py --table > $O/batch_sweep.json 2> $O/batch_sweep.txt; echo "sweep rc=$?"; cat $O/batch_sweep.txt
timeout 300 python tools/bench_sweep.py --group 64 --max 13 --ops encrypt,decrypt,mul --table > $O/batch_sweep_g64.json 2> $O/batch_sweep_g64.txt; echo "sweep g64 rc=$?"; cat $O/batch_sweep_g64.txt
timeout 300 python tools/bench_sweep.py --group 16 --max 14 --ops encrypt,decrypt,mul --table > $O/batch_sweep_g16.json 2> $O/batch_sweep_g16.txt; echo "sweep g16 rc=$?"; cat $O/batch_sweep_g16.txt
timeout 200 python tools/bench_sweep.py --min 16 --max 20 --ops pair_add,pair_add_rowb --table > $O/pair_rowb.json 2> $O/pair_rowb.txt; cat $O/pair_rowb.txt
timeout 900 python examples/benchmarks_batched.py --key-sizes 1024 2048 3072 4096 8192 > $O/benchmarks_batched.txt 2> $O/benchmarks_batched.err; echo "benchmarks rc=$?"; grep -v "^\[" $O/benchmarks_batched.txt
timeout 200 python-paillier_amd/lib/phe_microbench > $O/microbench.json 2> $O/microbench.err; echo "microbench rc=$?"
python - <<PY
import json
d=json.load(open("$O/microbench.json"))
for k in ("v_fma_f32","v_mad_u64_u32","v_mul_lo_u32","v_add_u32","v_lshl_add_u64","v_mov_b32_dpp_row_shl1","mfma_i32_16x16x64_i8","mix_8mad+1mfma_i8(mads)"):
    t=d["tests"][k]; print("%-32s %.3f cycles  %.3e lane-ops/s"%(k,t["cycles_per_wave_instr_per_simd"],t["lane_ops_per_s"]))
print(d["side_by_side_mfma_and_mad_waves"]); print(d.get("lone_wave_dependent_v_add_u32"))
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES -d $R/$O/prof_mb -- $R/python-paillier_amd/lib/phe_microbench > $R/$O/prof_mb.log 2>&1; echo "pmc microbench rc=$?"
cd $R
PHE_SUMMARIZE_ALL=1 python tools/rocprof_summarize.py $O/prof_mb > $O/microbench_pmc.txt 2>&1; rm -rf $O/prof_mb
grep -E "k_fma_f32|k_mad_u64_u32|k_side|k_clock|k_mfma" $O/microbench_pmc.txt | head -40
