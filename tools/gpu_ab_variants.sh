#!/bin/bash
# Same-box A/B of the measurement variants of the pair sweeps (tools/exp/build_variants.sh): interleaved
# bench runs, an occupancy sweep of the shipped kernel, and SQ_INSTS_VALU per variant.  UNITINV computes wrong values by
# design (timing only): bench.py exits 1 there, its line is still printed.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${TAG:-r02h}; mkdir -p $O; R=$PWD
L=$R/python-paillier_amd/lib
run() { # name lib extra-args
  PHE_HIP_LIB=$2 timeout 300 python bench.py --steps 3 --warmup 1 --no-ops --no-cpu-baseline --oracle-sample 64 $3 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('$1', round(d['value']), round(d['decrypt']['value']), d['bit_exact']['roundtrip_full_batch'], d['bit_exact']['strided_sample_vs_gmp_oracle'])" | tee -a $O/ab.txt
}
for round in 1 2; do
  run base $L/libphe_hip.so
  run qmad $L/libphe_hip_qmad.so
  run unitinv_timing_only $L/libphe_hip_unitinv.so
done
run base_blocks_per_cu_1 $L/libphe_hip.so "--blocks-per-cu 1"
run base_blocks_per_cu_2 $L/libphe_hip.so "--blocks-per-cu 2"
cd /tmp && export TMPDIR=/tmp
for v in "" _qmad _unitinv; do
  PHE_HIP_LIB=$L/libphe_hip$v.so timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE -d $R/$O/pmc$v -- python $R/bench.py --batch 131072 --steps 1 --warmup 0 --no-ops --no-cpu-baseline --oracle-sample 64 > /dev/null 2>&1
  echo "== variant '${v:-base}'" >> $R/$O/pmc.txt
  (cd $R && python tools/rocprof_summarize.py $O/pmc$v | grep -E "k_modexp_split" >> $O/pmc.txt)
  rm -rf $R/$O/pmc$v
done
cat $R/$O/pmc.txt
