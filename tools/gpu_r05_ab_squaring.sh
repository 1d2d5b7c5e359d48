#!/bin/bash
# Same-box A/B of the squaring sweep (round 5): shipped kernels (symmetric half of X0*X0, the doubled digit by one shift per row),
# PHE_VARIANT_SQROW (the doubled digit from a second LDS row instead of a shift per row), PHE_VARIANT_SQFULL (every product in both orders: the round-4 sweep).
# Variants: tools/exp/build_variants.sh SQFULL SQROW.  (profiles/r05c_ab_squaring.txt was made when the LDS row was the shipped form and
# the shift the variant: its labels say so.)   TAG=r05c bash tools/gpu_r05_ab_squaring.sh   (through gpurun)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/${TAG:-r05c}; mkdir -p $O; R=$PWD
L=$R/python-paillier_amd/lib
run() { # name lib
  PHE_HIP_LIB=$2 timeout 300 python bench.py --steps 3 --warmup 1 --no-ops --no-cpu-baseline --no-config4 --oracle-sample 64 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('$1', round(d['value']), round(d['decrypt']['value']), d['bit_exact']['roundtrip_full_batch'], d['bit_exact']['strided_sample_vs_gmp_oracle'])" | tee -a $O/ab_squaring.txt
}
for round in 1 2; do
  run shipped_symmetric_shift $L/libphe_hip.so
  run variant_symmetric_lds_row $L/libphe_hip_sqrow.so
  run variant_full_square_round4 $L/libphe_hip_sqfull.so
done
