#!/bin/bash
# does the ROCm ASan runtime (which interposes hsa_amd_memory_pool_allocate for its device allocator) let a HIP process start on this box?
cd "$GRAFT_REPO_ROOT" || exit 1
RT=$PWD/python-paillier_amd/lib/libclang_rt.asan-x86_64.so
export PHE_HIP_LIB=$PWD/python-paillier_amd/lib/libphe_hip_asan.so
try() { echo "== $1"; env $1 LD_PRELOAD=$RT timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4; }
try "ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 HSA_XNACK=1"
try "ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:allocator_may_return_null=1"
try "ASAN_OPTIONS=detect_leaks=0 HSA_XNACK=0"
try "ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:max_allocation_size_mb=65536 HSA_ENABLE_SDMA=0"
