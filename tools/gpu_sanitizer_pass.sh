#!/bin/bash
# TAG=<name> bash tools/gpu_sanitizer_pass.sh (through gpurun): the GPU parity + ladder tests over the sanitizer build
# (tools/build_sanitizer.sh: UBSan + libstdc++ assertions + stack protectors on the host side of the C-ABI, glibc heap checks, LDS
# bounds traps in the staging / wave-pair / table kernels).  MODE=asan tries the ASan + UBSan library instead (needs a host on
# which ROCm's ASan runtime can start a HIP process: not these boxes, see tools/build_sanitizer.sh).
cd "$GRAFT_REPO_ROOT" || exit 1
T=${TAG:-r04}; O=gpurun_out/$T; mkdir -p $O
if [ "$MODE" = asan ]; then
  RT=python-paillier_amd/lib/libclang_rt.asan-x86_64.so
  export PHE_HIP_LIB=$PWD/python-paillier_amd/lib/libphe_hip_asan.so
  export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1:protect_shadow_gap=0:detect_odr_violation=0:log_path=$PWD/$O/asan
else
  RT=""
  export PHE_HIP_LIB=$PWD/python-paillier_amd/lib/libphe_hip_san.so
  export LD_LIBRARY_PATH=$PWD/python-paillier_amd/lib:$LD_LIBRARY_PATH
  export MALLOC_CHECK_=3 MALLOC_PERTURB_=165
fi
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1:abort_on_error=1:log_path=$PWD/$O/ubsan
(time LD_PRELOAD=$RT timeout ${LIMIT:-540} python -m pytest tests/test_gpu_parity.py tests/test_gpu_ladder.py tests/test_abi_exports.py -q -m "gpu or not gpu" -x ${PYTEST_ARGS}) > $O/sanitizer_pytest.txt 2>&1
echo "pytest rc=$?" >> $O/sanitizer_pytest.txt
tail -15 $O/sanitizer_pytest.txt
ls $O | grep -E "^(asan|ubsan)" && head -60 $O/asan* $O/ubsan* 2>/dev/null | head -120
echo "sanitizer reports: $(ls $O | grep -cE '^(asan|ubsan)')"
