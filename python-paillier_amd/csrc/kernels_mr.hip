// kernels_mr.hip — batched Miller–Rabin with one modulus per row (csrc/primality.h) on 16-lane limb groups: the
// candidates of a prime search are few (hundreds), so the latency geometry is the right one — 16 rows per workgroup,
// S = 16 L limbs of 29 bits per row.
#include <hip/hip_runtime.h>
#include <stdint.h>

// clang-format off
#include "wave_gfx950.h"
#include "mont_core.h"
#include "primality.h"
// clang-format on

namespace phe {

constexpr int kMrBlock = 256;

template <int L>
__global__ void __launch_bounds__(kMrBlock) k_miller_rabin(MillerRabinArgs A) {
    constexpr int G = 16, S = G * L, kGroups = kMrBlock / G;
    __shared__ __attribute__((aligned(16))) uint32_t lds[kGroups * (S + kLdsPad)];
    const uint32_t grp = threadIdx.x / G;
    miller_rabin_body<G, L>(A, lds + grp * (S + kLdsPad), blockIdx.x * kGroups + grp, gridDim.x * kGroups, threadIdx.x & 63u);
}

namespace mr {

// L of the 16-lane geometry (key_setup.h kL16); -1 if this unit does not hold it
int launch(int L, int blocks, hipStream_t st, const MillerRabinArgs& A) {
    switch (L) {
#define X(LL) case LL: k_miller_rabin<LL><<<dim3(blocks), dim3(kMrBlock), 0, st>>>(A); return 0;
        X(1) X(2) X(3) X(5) X(7) X(9) X(14) X(18)
#undef X
        default: return -1;
    }
}

}  // namespace mr
}  // namespace phe
