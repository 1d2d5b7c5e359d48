// wave_gfx950.h — the CDNA4 (gfx950) wavefront primitives the multi-precision core is written against.
//
// A 64-lane wavefront is used as four independent 16-lane DPP rows; each row ("limb group")
// owns one big number whose limbs are blocked across the row's lanes (lane g of the row holds
// limbs [g*L, (g+1)*L)).  Every cross-lane movement the Montgomery core needs is a single-
// instruction DPP row operation; nothing here touches LDS or memory:
//
//   row_down1  v_mov_b32_dpp row_shl:1   bound_ctrl:0   lane g <- lane g+1   (top lane <- 0)
//   row_up1    v_mov_b32_dpp row_shr:1   bound_ctrl:0   lane g <- lane g-1   (lane 0  <- 0)
//   row_bcast0 v_mov_b32_dpp row_newbcast:0             lane g <- lane 0 of its row
//
// tests/emu/wave_emu.h provides the same names on the host (fibers) so tests can run
// mont_core.h on the CPU; tests/test_gpu_prims.py checks these semantics on the real GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PHE_DEV __device__ __forceinline__
#define PHE_LDS_PTR(T) T*

namespace wave {

constexpr int kRow = 16;  // lanes per limb group (= one DPP row)

PHE_DEV uint32_t lane_id() { return __lane_id(); }

// lane g <- lane g+1 within the 16-lane row; the row's top lane receives 0
PHE_DEV uint32_t row_down1(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x101 /*row_shl:1*/, 0xf, 0xf, true);
}
// lane g <- lane g-1 within the 16-lane row; the row's lane 0 receives 0
PHE_DEV uint32_t row_up1(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111 /*row_shr:1*/, 0xf, 0xf, true);
}
// every lane <- lane 0 of its own row
PHE_DEV uint32_t row_bcast0(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x150 /*row_newbcast:0*/, 0xf, 0xf, true);
}
// 64-bit lane mask of a per-lane predicate (SGPR pair)
PHE_DEV uint64_t ballot(bool p) { return __ballot(p); }

// orders this wave's LDS stores before its later LDS loads (lanes of one wave exchange
// operands through LDS; no inter-wave communication exists in these kernels)
PHE_DEV void lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 32x32+32 -> 64 multiply-add: v_mad_u64_u32 (cannot overflow: (2^32-1)^2 + 2^32-1 < 2^64)
PHE_DEV uint64_t mad(uint32_t a, uint32_t b, uint32_t c) { return (uint64_t)a * b + c; }
// add/sub with carry chains: v_add_co_u32 / v_addc_co_u32 / v_sub_co_u32 / v_subb_co_u32
PHE_DEV uint32_t addc(uint32_t a, uint32_t b, uint32_t cin, uint32_t& cout) {
    unsigned co;
    uint32_t r = __builtin_addc(a, b, cin, &co);
    cout = co;
    return r;
}
PHE_DEV uint32_t subb(uint32_t a, uint32_t b, uint32_t bin, uint32_t& bout) {
    unsigned bo;
    uint32_t r = __builtin_subc(a, b, bin, &bo);
    bout = bo;
    return r;
}

}  // namespace wave
