// wave_gfx950.h — the CDNA4 (gfx950) wavefront primitives the multi-precision core is written against.
//
// A 64-lane wavefront is used as independent limb groups of G = 16 lanes (one DPP row), G = 8 lanes
// (half a row), G = 4 lanes (one quad) or G = 2 lanes (half a quad); each group owns one big number whose limbs are blocked across its lanes
// (lane g of the group holds limbs [g*L, (g+1)*L)).  Every cross-lane movement the Montgomery core needs is a single-
// instruction DPP row operation; nothing here touches LDS or memory:
//
//   grp_down1  v_mov_b32_dpp row_shl:1   bound_ctrl:0   lane g <- lane g+1   (top lane <- 0)
//   grp_up1    v_mov_b32_dpp row_shr:1   bound_ctrl:0   lane g <- lane g-1   (lane 0  <- 0)
//   grp_bcast0 v_mov_b32_dpp row_newbcast:0             lane g <- lane 0 of its group
// (groups of 8 lanes add one v_and / a second DPP; see the templates below)
// G = 64, the whole wavefront as ONE group (the latency rung: one number per wave, see key_setup.h kS64), uses the
// wavefront-wide DPP shifts of the GFX9 family and a scalar broadcast:
//   grp_down1  v_mov_b32_dpp wave_shl:1  bound_ctrl:0   lane i <- lane i+1   (lane 63 <- 0)
//   grp_up1    v_mov_b32_dpp wave_shr:1  bound_ctrl:0   lane i <- lane i-1   (lane 0  <- 0)
//   grp_bcast0 v_readfirstlane_b32                      every lane <- lane 0 (through an SGPR)
//
// tests/emu/wave_emu.h provides the same names on the host (fibers) so tests can run
// mont_core.h on the CPU; tests/test_gpu_parity.py::test_wave_primitives checks these semantics on the real GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PHE_DEV __device__ __forceinline__
#define PHE_LDS_PTR(T) T*
// -DPHE_DEBUG_BOUNDS (tools/build_sanitizer.sh: the debug build of the sanitizer pass): an index into an LDS area that leaves
// the area traps the wavefront (the process dies with a GPU fault: a test run over such a build fails loudly).  Off in the
// product build: the conditions are wave-uniform compares in the hottest loops.
#if defined(PHE_DEBUG_BOUNDS)
#define PHE_BOUNDS(...)                      \
    do {                                     \
        if (!(__VA_ARGS__)) __builtin_trap(); \
    } while (0)
#else
#define PHE_BOUNDS(...) ((void)0)
#endif

namespace wave {

constexpr int kRow = 16;  // lanes of one DPP row

PHE_DEV uint32_t lane_id() { return __lane_id(); }

// Per-lane constants of a limb group of G lanes (G = 16: one DPP row; G = 8: half a row; G = 4: one quad;
// G = 2: half a quad).
template <int G>
struct Lanes {
    uint32_t lane;      // 0..63
    uint32_t g;         // position inside the group, 0..G-1
    uint32_t not_top;   // all-ones unless this is the group's top lane    (G < 16 only)
    uint32_t not_low;   // all-ones unless this is the group's lane 0      (G < 16 only)
    PHE_DEV explicit Lanes(uint32_t lane_) : lane(lane_), g(lane_ & (G - 1)) {
        not_top = (g == G - 1) ? 0u : 0xffffffffu;
        not_low = (g == 0) ? 0u : 0xffffffffu;
        // keep them plain data: "x & mask" is then one full-rate v_and_b32 instead of a half-rate v_cndmask_b32
        // on an SGPR-pair condition (profiles/microbench_r01.json)
        asm volatile("" : "+v"(not_top), "+v"(not_low));
    }
};

PHE_DEV uint32_t dpp_row_shl1(uint32_t x) {  // lane i <- lane i+1 in the 16-lane row, lane 15 <- 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x101 /*row_shl:1*/, 0xf, 0xf, true);
}
PHE_DEV uint32_t dpp_row_shr1(uint32_t x) {  // lane i <- lane i-1 in the 16-lane row, lane 0 <- 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111 /*row_shr:1*/, 0xf, 0xf, true);
}

// lane g <- lane g+1 of its group; the group's top lane receives 0
// (G = 1, one number per lane: there is no neighbour — "the lane above" hands down 0, lane 0 of the group is the lane itself)
template <int G>
PHE_DEV uint32_t grp_down1(uint32_t x, const Lanes<G>& l) {
    if constexpr (G == 1) return 0u;
    else if constexpr (G == 64) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x130 /*wave_shl:1*/, 0xf, 0xf, true);
    else if constexpr (G == 16) return dpp_row_shl1(x);
    else if constexpr (G == 8) return dpp_row_shl1(x) & l.not_top;
    else if constexpr (G == 4) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xF9 /*quad_perm:[1,2,3,3]*/, 0xf, 0xf, true) & l.not_top;
    else return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xF5 /*quad_perm:[1,1,3,3]*/, 0xf, 0xf, true) & l.not_top;
}
// lane g <- lane g+1 of its group, nothing guaranteed for the group's top lane (G = 16: 0): for callers that
// fold "& not_top" into a mask they apply anyway
template <int G>
PHE_DEV uint32_t grp_down1_raw(uint32_t x) {
    if constexpr (G == 1) return x;  // (the caller's mask, 0 in a group's top lane, clears it)
    else if constexpr (G == 64) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x130 /*wave_shl:1*/, 0xf, 0xf, true);
    else if constexpr (G >= 8) return dpp_row_shl1(x);
    else if constexpr (G == 4) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xF9 /*quad_perm:[1,2,3,3]*/, 0xf, 0xf, true);
    else return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xF5 /*quad_perm:[1,1,3,3]*/, 0xf, 0xf, true);
}
// lane g <- lane g-1 of its group; the group's lane 0 receives 0
template <int G>
PHE_DEV uint32_t grp_up1(uint32_t x, const Lanes<G>& l) {
    if constexpr (G == 1) return 0u;
    else if constexpr (G == 64) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138 /*wave_shr:1*/, 0xf, 0xf, true);
    else if constexpr (G == 16) return dpp_row_shr1(x);
    else if constexpr (G == 8) return dpp_row_shr1(x) & l.not_low;
    else if constexpr (G == 4) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x90 /*quad_perm:[0,0,1,2]*/, 0xf, 0xf, true) & l.not_low;
    else return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xA0 /*quad_perm:[0,0,2,2]*/, 0xf, 0xf, true) & l.not_low;
}
// every lane <- lane 0 of its group
template <int G>
PHE_DEV uint32_t grp_bcast0(uint32_t x, const Lanes<G>&) {
    if constexpr (G == 1) {
        return x;
    } else if constexpr (G == 64) {
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)x);  // all 64 lanes are active in these kernels
    } else if constexpr (G == 16) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x150 /*row_newbcast:0*/, 0xf, 0xf, true);
    } else if constexpr (G == 4) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x00 /*quad_perm:[0,0,0,0]*/, 0xf, 0xf, true);
    } else if constexpr (G == 2) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xA0 /*quad_perm:[0,0,2,2]*/, 0xf, 0xf, true);
    } else {
        // quad_perm:[0,0,0,0] puts lanes 0,4,8,12 into their quads; row_shr:4 restricted to banks 1 and 3
        // then copies quad 0 -> quad 1 and quad 2 -> quad 3, i.e. lanes 0 and 8 to their 8-lane groups
        const int q = __builtin_amdgcn_update_dpp(0, (int)x, 0x00 /*quad_perm:[0,0,0,0]*/, 0xf, 0xf, true);
        return (uint32_t)__builtin_amdgcn_update_dpp(q, q, 0x114 /*row_shr:4*/, 0xf, 0xa, false);
    }
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

// Barrier of the whole workgroup (s_barrier), ordering LDS and global stores before it against loads after it for the waves
// of the group.  Only the two-waves-per-number kernels use it (split_core.h "one number on two wavefronts"); every other
// kernel here keeps its waves independent.
PHE_DEV void block_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// OR into a word of LDS that other waves of the workgroup OR into as well (ds_or_b32; ordered by the next block_barrier)
PHE_DEV void lds_or(uint32_t* p, uint32_t v) {
    __hip_atomic_fetch_or((__attribute__((address_space(3))) uint32_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Asynchronous 16-byte copy global -> LDS without a register round trip (global_load_lds_dwordx4, "LDS-DMA"): lane l
// of the wave lands at lds_wave_base + 16*l bytes — the destination is wave-uniform base + lane*16, the SOURCE is per
// lane.  Lanes for which `active` is false copy nothing.  The data may be read after wait_async_copies().
PHE_DEV void async_copy16_to_lds(const uint32_t* gsrc, uint32_t* lds_wave_base, bool active) {
    if (active)
        __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
PHE_DEV void wait_async_copies() {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0); expcnt / lgkmcnt untouched (gfx9 encoding)
    lds_fence();
}

// A value the optimiser must treat as freshly produced.  Every memory / LDS offset of the operand movement code is a
// function of the lane's position in its group, i.e. invariant across the element loop of a kernel: left alone, the
// compiler hoists hundreds of precomputed offsets out of that loop, runs out of registers and spills them (round 2:
// k_mulmod 340 VGPRs = one wave per SIMD, the staged variant 90 spilled VGPRs).  Re-reading the position where such code
// starts makes it recompute its offsets — a few hundred full-rate instructions next to tens of thousands of multiply-adds.
PHE_DEV uint32_t reread(uint32_t x) {
    asm volatile("" : "+v"(x));
    return x;
}

// The same for a (wave-uniform) pointer to per-key constants: a load through it is not hoisted out of the element loop
// into registers that would stay live — and be spilled — across every product.
PHE_DEV const uint32_t* reread_ptr(const uint32_t* p) {
    asm volatile("" : "+s"(p));
    return p;
}

// ... and for a per-lane pointer (a group's window table, an element's exponent row)
template <class T>
PHE_DEV T* reread_vptr(T* p) {
    asm volatile("" : "+v"(p));
    return p;
}

// ... and for a 64-bit accumulator: what was added to it so far stays one value (the optimiser does not re-associate a chain
// of multiply-adds across this point) — the latency sweeps of split_core.h order their dependent chain by hand
PHE_DEV uint64_t reread64(uint64_t x) {
    asm volatile("" : "+v"(x));
    return x;
}

// A generic pointer that is known to point into LDS, as an LDS pointer: ds_* instructions where the compiler cannot see the
// address space itself (a per-lane choice between two LDS areas) instead of flat_* ones.
typedef __attribute__((address_space(3))) uint32_t lds_u32;
PHE_DEV lds_u32* as_lds(uint32_t* p) { return (lds_u32*)p; }
// ... re-read (see reread_vptr): the address is recomputed here, and stays an LDS address (ds_* with immediate offsets)
PHE_DEV lds_u32* reread_lds(uint32_t* p) {
    lds_u32* q = (lds_u32*)p;
    asm volatile("" : "+v"(q));
    return q;
}
PHE_DEV const lds_u32* reread_lds(const uint32_t* p) {
    const lds_u32* q = (const lds_u32*)p;
    asm volatile("" : "+v"(q));
    return q;
}
// two words to an 8-byte aligned LDS address as one ds_write_b64
PHE_DEV void lds_store2(lds_u32* p, uint32_t a, uint32_t b) {
    typedef uint32_t __attribute__((ext_vector_type(2))) u32x2;
    *(__attribute__((address_space(3))) u32x2*)p = (u32x2){a, b};
}
// four words to a 16-byte aligned LDS address as one ds_write_b128
PHE_DEV void lds_store4(lds_u32* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
    *(__attribute__((address_space(3))) u32x4*)p = (u32x4){a, b, c, d};
}

// A wave-uniform value as the compiler should see it (an SGPR): v_readfirstlane
PHE_DEV uint32_t uniform(uint32_t x) { return __builtin_amdgcn_readfirstlane(x); }

// N consecutive words of batch-constant data at a WAVE-UNIFORM address, through the scalar data cache (s_load_dwordx*) into
// SGPRs: a multiply-add then takes the word as its scalar operand — one 4-byte read feeds all 64 lanes, no LDS or VGPR traffic
// (mul_tile.h: lane = element, the fold table word is the same for every lane).  The constant address space tells the
// compiler the memory is never written by this kernel, which is what lets it use the scalar path next to the kernel's stores.
template <int N>
PHE_DEV void scalar_words(uint32_t (&c)[N], const uint32_t* p) {
    typedef const uint32_t __attribute__((address_space(4))) const_u32;
    const_u32* q = (const_u32*)(uintptr_t)p;
#pragma unroll
    for (int k = 0; k < N; ++k) c[k] = q[k];
}

// 16 bytes of global memory at a 4-byte aligned address as one global_load_dwordx4
PHE_DEV void load16(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d, const uint32_t* p) {
    typedef uint32_t __attribute__((ext_vector_type(4), aligned(4))) u32x4_a4;
    const u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(p);
    a = v[0];
    b = v[1];
    c = v[2];
    d = v[3];
}
// Issue priority of this wave among the waves of its SIMD (0 lowest ... 3; the arbiter takes the highest priority first, the
// oldest wave among equals).  The waves of a workgroup that share a SIMD are served oldest first: with equal work the youngest
// finishes last and alone — and a lone wave leaves many of the SIMD's issue slots empty (a chain-free multiply-add loop: 0.38 of the
// nominal rate from one wave per SIMD, 0.68 from two).  A wave that lowers its priority as it gets
// through its share lets the ones behind it catch up (mul_tile.h).
PHE_DEV void set_priority(int level) {
    switch (level) {
        case 3: __builtin_amdgcn_s_setprio(3); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        default: __builtin_amdgcn_s_setprio(0); break;
    }
}
// the instruction scheduler moves nothing across this point
PHE_DEV void order_fence() { __builtin_amdgcn_sched_barrier(0); }

// ---- table words on the scalar path, requested ahead of their use (mul_tile.h) -------------------------------------------------
// ScalarRow<N>: N (5, 9, 10, 14 or 18: one to three s_load of 1 ... 16 words) consecutive words at a wave-uniform address, in SGPRs.  request() issues the s_load and
// returns at once; the words may be read after arrived() — the s_waitcnt is written by hand because the compiler, left to
// place the loads itself, sinks them to their first use and waits there (scalar loads return out of order: the only wait
// there is waits for every outstanding one, so a wait in front of a use must come BEFORE the next requests are issued).
typedef uint32_t __attribute__((ext_vector_type(16))) u32x16;
typedef uint32_t __attribute__((ext_vector_type(8))) u32x8;
typedef uint32_t __attribute__((ext_vector_type(2))) u32x2;
typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
template <int N>
struct ScalarRow;
template <>
struct ScalarRow<18> {
    u32x16 lo;
    u32x2 hi;
    PHE_DEV void request(const uint32_t* p) {
        asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx2 %1, %2, 0x40" : "=&s"(lo), "=&s"(hi) : "s"(p) : "memory");
    }
    template <int OFFW>  // the row OFFW words past p: an immediate of the load, no pointer arithmetic on the scalar unit
    PHE_DEV void request_at(const uint32_t* p) {
        asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx2 %1, %2, %4" : "=&s"(lo), "=&s"(hi) : "s"(p), "n"(4 * OFFW), "n"(4 * OFFW + 0x40) : "memory");
    }
    PHE_DEV uint32_t word(int k) const { return k < 16 ? lo[k] : hi[k - 16]; }
    PHE_DEV void landed() { asm volatile("" : "+s"(lo), "+s"(hi)); }
};
template <>
struct ScalarRow<10> {
    u32x8 lo;
    u32x2 hi;
    PHE_DEV void request(const uint32_t* p) {
        asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx2 %1, %2, 0x20" : "=&s"(lo), "=&s"(hi) : "s"(p) : "memory");
    }
    template <int OFFW>
    PHE_DEV void request_at(const uint32_t* p) {
        asm volatile("s_load_dwordx8 %0, %2, %3\n\ts_load_dwordx2 %1, %2, %4" : "=&s"(lo), "=&s"(hi) : "s"(p), "n"(4 * OFFW), "n"(4 * OFFW + 0x20) : "memory");
    }
    PHE_DEV uint32_t word(int k) const { return k < 8 ? lo[k] : hi[k - 8]; }
    PHE_DEV void landed() { asm volatile("" : "+s"(lo), "+s"(hi)); }
};
template <>
struct ScalarRow<14> {
    u32x8 lo;
    u32x4 mid;
    u32x2 hi;
    PHE_DEV void request(const uint32_t* p) {
        asm volatile("s_load_dwordx8 %0, %3, 0x0\n\ts_load_dwordx4 %1, %3, 0x20\n\ts_load_dwordx2 %2, %3, 0x30"
                     : "=&s"(lo), "=&s"(mid), "=&s"(hi)
                     : "s"(p)
                     : "memory");
    }
    template <int OFFW>
    PHE_DEV void request_at(const uint32_t* p) {
        asm volatile("s_load_dwordx8 %0, %3, %4\n\ts_load_dwordx4 %1, %3, %5\n\ts_load_dwordx2 %2, %3, %6"
                     : "=&s"(lo), "=&s"(mid), "=&s"(hi)
                     : "s"(p), "n"(4 * OFFW), "n"(4 * OFFW + 0x20), "n"(4 * OFFW + 0x30)
                     : "memory");
    }
    PHE_DEV uint32_t word(int k) const { return k < 8 ? lo[k] : (k < 12 ? mid[k - 8] : hi[k - 12]); }
    PHE_DEV void landed() { asm volatile("" : "+s"(lo), "+s"(mid), "+s"(hi)); }
};
template <>
struct ScalarRow<9> {
    u32x8 lo;
    uint32_t hi;
    PHE_DEV void request(const uint32_t* p) {
        asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dword %1, %2, 0x20" : "=&s"(lo), "=&s"(hi) : "s"(p) : "memory");
    }
    template <int OFFW>
    PHE_DEV void request_at(const uint32_t* p) {
        asm volatile("s_load_dwordx8 %0, %2, %3\n\ts_load_dword %1, %2, %4" : "=&s"(lo), "=&s"(hi) : "s"(p), "n"(4 * OFFW), "n"(4 * OFFW + 0x20) : "memory");
    }
    PHE_DEV uint32_t word(int k) const { return k < 8 ? lo[k] : hi; }
    PHE_DEV void landed() { asm volatile("" : "+s"(lo), "+s"(hi)); }
};
template <>
struct ScalarRow<5> {
    u32x4 lo;
    uint32_t hi;
    PHE_DEV void request(const uint32_t* p) {
        asm volatile("s_load_dwordx4 %0, %2, 0x0\n\ts_load_dword %1, %2, 0x10" : "=&s"(lo), "=&s"(hi) : "s"(p) : "memory");
    }
    template <int OFFW>
    PHE_DEV void request_at(const uint32_t* p) {
        asm volatile("s_load_dwordx4 %0, %2, %3\n\ts_load_dword %1, %2, %4" : "=&s"(lo), "=&s"(hi) : "s"(p), "n"(4 * OFFW), "n"(4 * OFFW + 0x10) : "memory");
    }
    PHE_DEV uint32_t word(int k) const { return k < 4 ? lo[k] : hi; }
    PHE_DEV void landed() { asm volatile("" : "+s"(lo), "+s"(hi)); }
};
// Two LDS words 64 rows of 4 bytes apart (rows ROW and ROW + 1 of a [row][64 lanes] buffer, this lane's column), requested the
// same way: were the read left to the compiler, its own wait for it — the counter is shared with the scalar loads — would sit
// in front of the first use and wait for the requests issued since.
struct DigitPair {
    u32x2 v;
    template <int ROW>
    PHE_DEV void request(const uint32_t* lds_column) {
        const uint32_t addr = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t*)lds_column;
        asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=&v"(v) : "v"(addr), "n"(ROW), "n"(ROW + 1) : "memory");
    }
    PHE_DEV uint32_t word(int u) const { return v[u]; }
    PHE_DEV void landed() { asm volatile("" : "+v"(v)); }
};
// every request issued so far has landed (and every LDS read: the counter is shared); c / d: the group read next (handing
// them through here is what orders their uses after the wait)
template <int N, int GD>
PHE_DEV void arrived(ScalarRow<N> (&c)[GD], DigitPair (&d)[GD / 2]) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < GD; ++u) c[u].landed();  // (volatile, so they stay behind the wait; the uses depend on what they hand back)
#pragma unroll
    for (int u = 0; u < GD / 2; ++u) d[u].landed();
}

// 32x32+64 -> 64 multiply-accumulate: v_mad_u64_u32 with a full 64-bit addend.  The radix-2^29 core
// keeps every accumulator below 2^64 by construction, so the carry-out is never needed.
PHE_DEV uint64_t mad64(uint32_t a, uint32_t b, uint64_t c) { return (uint64_t)a * b + c; }

}  // namespace wave
