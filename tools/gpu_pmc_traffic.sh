#!/bin/bash
# TAG=<name> bash tools/gpu_pmc_traffic.sh (through gpurun): the PMC passes behind bench.py's roofline.traffic / pmc_valu figures on the
# CURRENT tree -> gpurun_out/$TAG/hbm_traffic.json (copy to profiles/hbm_traffic_rNN.json).  Own runs, counters only (no tracing).
cd "$GRAFT_REPO_ROOT" || exit 1
T=${TAG:-r04}; O=gpurun_out/$T; mkdir -p $O; R=$PWD; B=${BATCH:-131072}
cd /tmp && export TMPDIR=/tmp
for leg in encrypt decrypt; do
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS -d $R/$O/pmc_${leg}_valu -- python $R/bench.py --batch $B --steps 1 --warmup 0 --only $leg > $R/$O/pmc_${leg}_valu.log 2>&1; echo "$leg valu rc=$?"
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/$O/pmc_${leg}_fetch -- python $R/bench.py --batch $B --steps 1 --warmup 0 --only $leg > $R/$O/pmc_${leg}_fetch.log 2>&1; echo "$leg fetch rc=$?"
  timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/$O/pmc_${leg}_write -- python $R/bench.py --batch $B --steps 1 --warmup 0 --only $leg > $R/$O/pmc_${leg}_write.log 2>&1; echo "$leg write rc=$?"
done
# the kernels behind bench.py's `ops` object (configs[2]; VERDICT round 4 missing 6): k_mulmod_tile<9> / <14>, k_modexp_var_split, k_pair_mul,
# k_to_pair — tools/bench_sweep.py at ONE batch size ($B rows per dispatch), the same three passes
OPS_LEGS=""
for bits in ${OPS_BITS:-1024 2048 3072}; do
  sweep="python $R/tools/bench_sweep.py --key-bits $bits --min 17 --max 17 --ops add,mul,pair_add --budget-ms 30"
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS -d $R/$O/pmc_ops${bits}_valu -- $sweep > $R/$O/pmc_ops${bits}_valu.log 2>&1; echo "ops $bits valu rc=$?"
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/$O/pmc_ops${bits}_fetch -- $sweep > $R/$O/pmc_ops${bits}_fetch.log 2>&1; echo "ops $bits fetch rc=$?"
  timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/$O/pmc_ops${bits}_write -- $sweep > $R/$O/pmc_ops${bits}_write.log 2>&1; echo "ops $bits write rc=$?"
  OPS_LEGS="$OPS_LEGS --leg ops$bits $O/pmc_ops${bits}_valu $O/pmc_ops${bits}_fetch $O/pmc_ops${bits}_write"
done
cd $R
python tools/pmc_traffic_json.py $OPS_LEGS --batch $B --source "rocprofv3 --pmc on the tree of $T (tools/gpu_pmc_traffic.sh: bench.py --only <leg> --batch $B --steps 1 --warmup 0)" \
  --leg encrypt $O/pmc_encrypt_valu $O/pmc_encrypt_fetch $O/pmc_encrypt_write --leg decrypt $O/pmc_decrypt_valu $O/pmc_decrypt_fetch $O/pmc_decrypt_write > $O/hbm_traffic.json
python tools/rocprof_summarize.py $O/pmc_encrypt_valu $O/pmc_encrypt_fetch $O/pmc_encrypt_write $O/pmc_decrypt_valu $O/pmc_decrypt_fetch $O/pmc_decrypt_write $O/pmc_ops*_valu $O/pmc_ops*_fetch $O/pmc_ops*_write > $O/rocprofv3_pmc.txt 2>&1
rm -rf $O/pmc_encrypt_valu $O/pmc_encrypt_fetch $O/pmc_encrypt_write $O/pmc_decrypt_valu $O/pmc_decrypt_fetch $O/pmc_decrypt_write $O/pmc_ops*_valu $O/pmc_ops*_fetch $O/pmc_ops*_write
cat $O/hbm_traffic.json | head -40
