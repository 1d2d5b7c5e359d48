#!/bin/bash
# The sanitizer build of the C-ABI: python-paillier_amd/lib/libphe_hip_san.so =
#   * phe_hip.hip — the 2,800 lines of raw-pointer HOST code behind the ABI (contexts, grow-only scratch, pinned / mapped staging,
#     the three-stream chunk pipeline) — compiled with UndefinedBehaviorSanitizer (bounds of arrays with known size, alignment,
#     shifts, signed overflow, null / misaligned pointer use, pointer overflow ...), libstdc++'s container assertions
#     (-D_GLIBCXX_ASSERTIONS: every std::vector / std::string index is range-checked) and stack protectors; the run adds glibc's
#     heap consistency checks (MALLOC_CHECK_=3, MALLOC_PERTURB_).
#     AddressSanitizer itself cannot run a HIP process on these boxes: ROCm's ASan runtime interposes
#     hsa_amd_memory_pool_allocate with its own device allocator and dies on the first pool allocation ("out-of-memory ...
#     0x400000 bytes", with HSA_XNACK=0 and =1: profiles/r04i_asan_runtime_cannot_start_hip.txt); the ASan+UBSan library is still
#     built (libphe_hip_asan.so) for a host where that runtime works;
#   * the kernel units whose LDS indexing is the most intricate (products through LDS-DMA staging, the table product, wave pairs,
#     the late sweeps) rebuilt with -DPHE_DEBUG_BOUNDS: an LDS index that leaves its area traps the wavefront;
#   * every other object as the product build made it.
# Run the GPU tests over it with tools/gpu_sanitizer_pass.sh.  Cross-compiles without a GPU.
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD; CSRC=$ROOT/python-paillier_amd/csrc; OBJ=$ROOT/build/obj; SAN=$ROOT/build/obj_san; mkdir -p $SAN
python -c "import __graft_entry__ as g; g.build_hip()"          # the product objects first
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -mllvm -pragma-unroll-threshold=1000000 -fPIC"
( cd $SAN && /opt/rocm/bin/hipcc $FLAGS -fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan -fno-omit-frame-pointer -c -o phe_hip.o $CSRC/phe_hip.hip ) &
( cd $SAN && /opt/rocm/bin/hipcc $FLAGS -fsanitize=undefined -fno-sanitize-recover=undefined -shared-libsan -fno-omit-frame-pointer -D_GLIBCXX_ASSERTIONS -fstack-protector-strong -c -o phe_hip_ub.o $CSRC/phe_hip.hip ) &
for u in kernels_g8a kernels_g8b kernels_g8c kernels_t16 kernels_s64a; do
  ( cd $SAN && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=1000000 -fPIC -DPHE_DEBUG_BOUNDS -c -o $u.o $CSRC/$u.hip ) &
done
wait
OBJS=""; OBJS_UB=""
for o in $OBJ/*.o; do
  b=$(basename $o)
  if [ -f $SAN/$b ]; then OBJS="$OBJS $SAN/$b"; else OBJS="$OBJS $o"; fi
  if [ $b = phe_hip.o ]; then OBJS_UB="$OBJS_UB $SAN/phe_hip_ub.o"; elif [ -f $SAN/$b ]; then OBJS_UB="$OBJS_UB $SAN/$b"; else OBJS_UB="$OBJS_UB $o"; fi
done
LIBD=$ROOT/python-paillier_amd/lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan -o $LIBD/libphe_hip_asan.so $OBJS
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=undefined -shared-libsan -Wl,-rpath,'$ORIGIN' -o $LIBD/libphe_hip_san.so $OBJS_UB
for rt in libclang_rt.asan-x86_64.so libclang_rt.ubsan_standalone-x86_64.so; do cp -f $(/opt/rocm/lib/llvm/bin/clang -print-file-name=$rt) $LIBD/ 2>/dev/null || true; done
ls -la $LIBD/libphe_hip_asan.so $LIBD/libphe_hip_san.so
