// microbench.hip — calibration of the gfx950 integer-VALU roofline this engine is priced against.
//
// SURVEY.md 8(d) assumes v_mad_u64_u32 is a quarter-rate op (peak 9.83e12 MAC32/s at 256 CUs,
// 2.4 GHz).  This standalone tool measures the sustained issue rate of the instructions the
// Montgomery kernels are made of (and of a few alternatives: 24-bit multiplies, f64 FMA), so that
// bench.py's roofline.peak is a measured number.  Every test runs 8 independent dependency chains
// per lane, 8 waves per SIMD, all CUs; results are printed as one JSON object.
// Round 3: a loop trip now issues 64 instructions per lane (the 8 chains, 8 times over) instead of 8 — the three scalar
// instructions and the taken branch of every trip cost ~6 cycles that 8 waves per SIMD did not hide (v_fma_f32 read 2.75
// cycles per wave-instruction against the 2 of the 157 TFLOP/s fp32 peak; the same constant sat on every other row) —
// and the report carries the cycle counts at the nominal AND at the measured shader clock.  New: an MFMA-only wave beside
// a multiply-add-only wave on the same SIMD (k_side_by_side), the experiment that decides whether the matrix cores can
// take work off the integer VALU.
//
//   hipcc --offload-arch=gfx950 -O3 microbench.hip -o phe_microbench && ./phe_microbench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <string>
#include <vector>

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            return 1;                                                             \
        }                                                                         \
    } while (0)

constexpr int kIters = 16384;  // instructions per chain; a loop trip issues kRepeat of them on each of the 8 chains
constexpr int kRepeat = 8;
constexpr int kChains = 8;

#define KERNEL_BEGIN(NAME)                                                         \
    __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {    \
        const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;                \
        uint32_t a = seed * 2654435761u + tid, b = (seed ^ tid) | 1u;              \
        (void)a; (void)b;

#define KERNEL_END(RESULT)                                                         \
        if ((RESULT) == 0x12345678u) out[tid] = (RESULT);                          \
    }

// --- 32-bit accumulator tests -----------------------------------------------------------------
#define DEF_U32_TEST(NAME, ASM)                                                    \
    KERNEL_BEGIN(NAME)                                                             \
        uint32_t x[kChains];                                                       \
        for (int i = 0; i < kChains; ++i) x[i] = a + i;                            \
        for (int it = 0; it < kIters / kRepeat; ++it) {                            \
            _Pragma("unroll") for (int rep = 0; rep < kRepeat; ++rep)              \
            _Pragma("unroll") for (int i = 0; i < kChains; ++i)                    \
                asm volatile(ASM : "+v"(x[i]) : "v"(a), "v"(b) : "vcc", "s20", "s21");  \
        }                                                                          \
        uint32_t r = 0;                                                            \
        for (int i = 0; i < kChains; ++i) r ^= x[i];                               \
    KERNEL_END(r)

DEF_U32_TEST(k_add_u32, "v_add_u32 %0, %0, %1")
DEF_U32_TEST(k_mov_b32, "v_mov_b32 %0, %1")
DEF_U32_TEST(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %2")
DEF_U32_TEST(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %2")
DEF_U32_TEST(k_mad_u32_u24, "v_mad_u32_u24 %0, %1, %2, %0")
DEF_U32_TEST(k_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %0, %2")
DEF_U32_TEST(k_fma_f32, "v_fma_f32 %0, %1, %2, %0")
DEF_U32_TEST(k_fmac_f32, "v_fmac_f32 %0, %1, %2")   // the same operation in the 4-byte VOP2 encoding (half the instruction bytes)
DEF_U32_TEST(k_dpp_row_shl1, "v_mov_b32_dpp %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0")
DEF_U32_TEST(k_dpp_newbcast, "v_mov_b32_dpp %0, %0 row_newbcast:0 row_mask:0xf bank_mask:0xf")
DEF_U32_TEST(k_addc_chain, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
DEF_U32_TEST(k_add_co_addc_pair, "v_add_co_u32 %0, s[20:21], %0, %1\n\tv_addc_co_u32 %0, s[20:21], %0, %2, s[20:21]")
DEF_U32_TEST(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEF_U32_TEST(k_cndmask_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
DEF_U32_TEST(k_and_b32, "v_and_b32 %0, %0, %1")
DEF_U32_TEST(k_alignbit, "v_alignbit_b32 %0, %0, %1, 30")

// --- 64-bit accumulator tests -----------------------------------------------------------------
#define DEF_U64_TEST(NAME, ASM)                                                    \
    KERNEL_BEGIN(NAME)                                                             \
        uint64_t x[kChains];                                                       \
        for (int i = 0; i < kChains; ++i) x[i] = ((uint64_t)b << 32) | (a + i);    \
        for (int it = 0; it < kIters / kRepeat; ++it) {                            \
            _Pragma("unroll") for (int rep = 0; rep < kRepeat; ++rep)              \
            _Pragma("unroll") for (int i = 0; i < kChains; ++i)                    \
                asm volatile(ASM : "+v"(x[i]) : "v"(a), "v"(b) : "vcc");           \
        }                                                                          \
        uint64_t r = 0;                                                            \
        for (int i = 0; i < kChains; ++i) r ^= x[i];                               \
    KERNEL_END((uint32_t)(r ^ (r >> 32)))

DEF_U64_TEST(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
DEF_U64_TEST(k_fma_f64, "v_fma_f64 %0, %0, %0, %0")
DEF_U64_TEST(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 1, %0")
DEF_U64_TEST(k_lshrrev_b64, "v_lshrrev_b64 %0, 1, %0")

// --- v_mad_u64_u32 with hand-placed registers: does the VGPR bank (index mod 4) of the operands matter? ---------
// 8 chains, accumulators ACC0..ACC7 (even-aligned pairs), multiplicands in SRC0 / SRC1; the whole loop is one asm block
// so the register numbers are exactly the ones written here.
#define DEF_MAD_PLACED(NAME, S0, S1, A0, A1, A2, A3, A4, A5, A6, A7)                                           \
    KERNEL_BEGIN(NAME)                                                                                          \
        uint32_t r;                                                                                             \
        asm volatile(                                                                                           \
            "v_mov_b32 v" #S0 ", %1\n\tv_mov_b32 v" #S1 ", %2\n\t"                                              \
            "v_mov_b32 v" #A0 ", %1\n\tv_mov_b32 v" #A1 ", %2\n\tv_mov_b32 v" #A2 ", %1\n\tv_mov_b32 v" #A3 ", %2\n\t" \
            "v_mov_b32 v" #A4 ", %1\n\tv_mov_b32 v" #A5 ", %2\n\tv_mov_b32 v" #A6 ", %1\n\tv_mov_b32 v" #A7 ", %2\n\t" \
            "s_movk_i32 s20, 0x4000\n"                                                                          \
            "1:\n\t"                                                                                            \
            "v_mad_u64_u32 v[" #A0 ":" #A0 "+1], vcc, v" #S0 ", v" #S1 ", v[" #A0 ":" #A0 "+1]\n\t"               \
            "v_mad_u64_u32 v[" #A1 ":" #A1 "+1], vcc, v" #S0 ", v" #S1 ", v[" #A1 ":" #A1 "+1]\n\t"               \
            "v_mad_u64_u32 v[" #A2 ":" #A2 "+1], vcc, v" #S0 ", v" #S1 ", v[" #A2 ":" #A2 "+1]\n\t"               \
            "v_mad_u64_u32 v[" #A3 ":" #A3 "+1], vcc, v" #S0 ", v" #S1 ", v[" #A3 ":" #A3 "+1]\n\t"               \
            "v_mad_u64_u32 v[" #A4 ":" #A4 "+1], vcc, v" #S0 ", v" #S1 ", v[" #A4 ":" #A4 "+1]\n\t"               \
            "v_mad_u64_u32 v[" #A5 ":" #A5 "+1], vcc, v" #S0 ", v" #S1 ", v[" #A5 ":" #A5 "+1]\n\t"               \
            "v_mad_u64_u32 v[" #A6 ":" #A6 "+1], vcc, v" #S0 ", v" #S1 ", v[" #A6 ":" #A6 "+1]\n\t"               \
            "v_mad_u64_u32 v[" #A7 ":" #A7 "+1], vcc, v" #S0 ", v" #S1 ", v[" #A7 ":" #A7 "+1]\n\t"               \
            "s_sub_u32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b\n\t"                             \
            "v_xor_b32 %0, v" #A0 ", v" #A1 "\n\tv_xor_b32 %0, %0, v" #A2 "\n\tv_xor_b32 %0, %0, v" #A3 "\n\t"    \
            "v_xor_b32 %0, %0, v" #A4 "\n\tv_xor_b32 %0, %0, v" #A5 "\n\tv_xor_b32 %0, %0, v" #A6 "\n\tv_xor_b32 %0, %0, v" #A7 \
            : "=&v"(r) : "v"(a), "v"(b)                                                                         \
            : "vcc", "scc", "s20", "v" #S0, "v" #S1, "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49",  \
              "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64",   \
              "v65", "v66", "v67", "v68", "v69", "v70", "v71");                                               \
    KERNEL_END(r)

// accumulators on banks (0,1), multiplicands on banks 2 and 3: no operand shares a bank
DEF_MAD_PLACED(k_mad_banks_apart, 34, 35, 40, 44, 48, 52, 56, 60, 64, 68)
// accumulators on banks (0,1), multiplicands on banks 0 and 1: both collide with the addend
DEF_MAD_PLACED(k_mad_banks_clash, 32, 33, 40, 44, 48, 52, 56, 60, 64, 68)
// multiplicands on the same bank as each other, accumulators elsewhere
DEF_MAD_PLACED(k_mad_srcs_same_bank, 34, 38, 40, 44, 48, 52, 56, 60, 64, 68)
// accumulators alternating (0,1) / (2,3), multiplicands on banks 2 and 3: what a compiler-chosen layout looks like
DEF_MAD_PLACED(k_mad_banks_mixed, 34, 35, 40, 42, 44, 46, 48, 50, 52, 54)

// --- the inner-loop mix of mont_core.h: mad + addc (+ mov), to see what co-issues ---------------
KERNEL_BEGIN(k_mix_mad_addc)
    uint64_t x[kChains];
    uint32_t y[kChains];
    for (int i = 0; i < kChains; ++i) { x[i] = a + i; y[i] = b + i; }
    for (int it = 0; it < kIters; ++it) {
        _Pragma("unroll") for (int i = 0; i < kChains; ++i) {
            asm volatile("v_mad_u64_u32 %0, s[20:21], %2, %3, %0\n\tv_addc_co_u32 %1, vcc, %1, %2, vcc"
                         : "+v"(x[i]), "+v"(y[i]) : "v"(a), "v"(b) : "vcc", "s20", "s21");
        }
    }
    uint64_t r = 0;
    for (int i = 0; i < kChains; ++i) r ^= x[i] ^ y[i];
KERNEL_END((uint32_t)(r ^ (r >> 32)))

KERNEL_BEGIN(k_mix_mad_addc_mov)
    uint64_t x[kChains];
    uint32_t y[kChains], z[kChains];
    for (int i = 0; i < kChains; ++i) { x[i] = a + i; y[i] = b + i; z[i] = i; }
    for (int it = 0; it < kIters; ++it) {
        _Pragma("unroll") for (int i = 0; i < kChains; ++i) {
            asm volatile("v_mad_u64_u32 %0, s[20:21], %3, %4, %0\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc\n\tv_mov_b32 %2, %4"
                         : "+v"(x[i]), "+v"(y[i]), "+v"(z[i]) : "v"(a), "v"(b) : "vcc", "s20", "s21");
        }
    }
    uint64_t r = 0;
    for (int i = 0; i < kChains; ++i) r ^= x[i] ^ y[i] ^ z[i];
KERNEL_END((uint32_t)(r ^ (r >> 32)))

// --- the matrix pipe beside the integer VALU: int8 MFMA alone, and one MFMA per 16 v_mad_u64_u32 ---------------
// (groundwork for moving the batch-constant half of the reductions, m*n, onto the otherwise idle matrix cores)
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_mfma_i8(uint32_t* out, uint32_t seed) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    v4i a = {(int)(seed + tid), (int)(seed ^ tid), (int)tid, (int)seed}, b = {(int)tid, 3, (int)seed, 7};
    v4i c[4] = {{0, 0, 0, 0}, {1, 1, 1, 1}, {2, 2, 2, 2}, {3, 3, 3, 3}};
    for (int it = 0; it < kIters / 4; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[i], 0, 0, 0);
    }
    const v4i r = c[0] + c[1] + c[2] + c[3];
    if ((uint32_t)(r.x ^ r.y ^ r.z ^ r.w) == 0x12345678u) out[tid] = (uint32_t)r.x;
}
__global__ void __launch_bounds__(256) k_mix_mfma_mad(uint32_t* out, uint32_t seed) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t aa = seed * 2654435761u + tid, bb = (seed ^ tid) | 1u;
    v4i a = {(int)(seed + tid), (int)(seed ^ tid), (int)tid, (int)seed}, b = {(int)tid, 3, (int)seed, 7};
    v4i c[2] = {{0, 0, 0, 0}, {1, 1, 1, 1}};
    uint64_t x[kChains];
    for (int i = 0; i < kChains; ++i) x[i] = ((uint64_t)bb << 32) | (aa + i);
    for (int it = 0; it < kIters / 2; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            c[h] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[h], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < kChains; ++i)
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[i]) : "v"(aa), "v"(bb) : "vcc");
        }
    }
    uint64_t r = 0;
    for (int i = 0; i < kChains; ++i) r ^= x[i];
    const v4i s = c[0] + c[1];
    if ((uint32_t)(r ^ (r >> 32)) + (uint32_t)(s.x ^ s.y ^ s.z ^ s.w) == 0x12345678u) out[tid] = (uint32_t)r;
}

// --- an MFMA-only wave BESIDE a multiply-add-only wave on one SIMD --------------------------------------------------
// 512-thread workgroups = 8 waves = two per SIMD.  `split` picks which waves are the matrix ones: 0 -> waves 4..7 (the
// second wave of every SIMD if waves go round the SIMDs in order), 1 -> the odd waves; both are run so that the answer
// does not hinge on the dispatch order.  mode bit 0: the multiply-add waves work, bit 1: the matrix waves work (an idle
// wave leaves at once).  If the two pipes overlap, mode 3 takes about as long as the longer of modes 1 and 2; if a SIMD
// issues one or the other, it takes their sum.
constexpr int kSideIters = 4096;
__global__ void __launch_bounds__(512) k_side_by_side(uint32_t* out, uint32_t seed, int mode, int split) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t wave = threadIdx.x >> 6;
    const bool matrix_wave = split ? (wave & 1u) : (wave >= 4u);
    if (matrix_wave) {
        if (!(mode & 2)) return;
        v4i a = {(int)(seed + tid), (int)(seed ^ tid), (int)tid, (int)seed}, b = {(int)tid, 3, (int)seed, 7};
        v4i c[4] = {{0, 0, 0, 0}, {1, 1, 1, 1}, {2, 2, 2, 2}, {3, 3, 3, 3}};
        for (int it = 0; it < kSideIters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[i], 0, 0, 0);
        }
        const v4i r = c[0] + c[1] + c[2] + c[3];
        if ((uint32_t)(r.x ^ r.y ^ r.z ^ r.w) == 0x12345678u) out[tid] = (uint32_t)r.x;
    } else {
        if (!(mode & 1)) return;
        const uint32_t aa = seed * 2654435761u + tid, bb = (seed ^ tid) | 1u;
        uint64_t x[kChains];
        for (int i = 0; i < kChains; ++i) x[i] = ((uint64_t)bb << 32) | (aa + i);
        for (int it = 0; it < kSideIters; ++it) {
#pragma unroll
            for (int rep = 0; rep < kRepeat; ++rep)
#pragma unroll
                for (int i = 0; i < kChains; ++i)
                    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[i]) : "v"(aa), "v"(bb) : "vcc");
        }
        uint64_t r = 0;
        for (int i = 0; i < kChains; ++i) r ^= x[i];
        if ((uint32_t)(r ^ (r >> 32)) == 0x12345678u) out[tid] = (uint32_t)r;
    }
}

// The same question with the VALU SATURATED: 768-thread workgroups = 12 waves = three per SIMD (waves go round the SIMDs:
// wave w runs on SIMD w % 4 — k_side_by_side's timings above are consistent with that and with nothing else); waves 0..3 are
// the matrix waves (one per SIMD), waves 4..11 the multiply-add waves (two per SIMD: enough to keep the VALU issuing
// back to back).  Besides the wall time, a multiply-add wave and a matrix wave each report the shader cycles (s_memtime)
// their own loop took, which no clock assumption enters.
__global__ void __launch_bounds__(768) k_side_by_side_saturated(uint32_t* out, uint32_t seed, int mode, unsigned long long* ticks) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t wave = threadIdx.x >> 6;
    if (wave < 4u) {
        if (!(mode & 2)) return;
        v4i a = {(int)(seed + tid), (int)(seed ^ tid), (int)tid, (int)seed}, b = {(int)tid, 3, (int)seed, 7};
        v4i c[4] = {{0, 0, 0, 0}, {1, 1, 1, 1}, {2, 2, 2, 2}, {3, 3, 3, 3}};
        const unsigned long long t0 = clock64();
        for (int it = 0; it < kSideIters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[i], 0, 0, 0);
        }
        const v4i r = c[0] + c[1] + c[2] + c[3];
        const unsigned long long t1 = clock64();
        if (blockIdx.x == 0 && threadIdx.x == 0) ticks[1] = t1 - t0;
        if ((uint32_t)(r.x ^ r.y ^ r.z ^ r.w) == 0x12345678u) out[tid] = (uint32_t)r.x;
    } else {
        if (!(mode & 1)) return;
        const uint32_t aa = seed * 2654435761u + tid, bb = (seed ^ tid) | 1u;
        uint64_t x[kChains];
        for (int i = 0; i < kChains; ++i) x[i] = ((uint64_t)bb << 32) | (aa + i);
        const unsigned long long t0 = clock64();
        for (int it = 0; it < kSideIters; ++it) {
#pragma unroll
            for (int rep = 0; rep < kRepeat; ++rep)
#pragma unroll
                for (int i = 0; i < kChains; ++i)
                    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[i]) : "v"(aa), "v"(bb) : "vcc");
        }
        uint64_t r = 0;
        for (int i = 0; i < kChains; ++i) r ^= x[i];
        const unsigned long long t1 = clock64();
        if (blockIdx.x == 0 && threadIdx.x == 256) ticks[0] = t1 - t0;
        if ((uint32_t)(r ^ (r >> 32)) == 0x12345678u) out[tid] = (uint32_t)r;
    }
}

// shader cycles one wave needs for its own stream of independent v_mad_u64_u32 when `blockDim / 256` waves share its SIMD
__global__ void __launch_bounds__(1024) k_mad_cycles(uint32_t* out, uint32_t seed, unsigned long long* ticks) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t aa = seed * 2654435761u + tid, bb = (seed ^ tid) | 1u;
    uint64_t x[kChains];
    for (int i = 0; i < kChains; ++i) x[i] = ((uint64_t)bb << 32) | (aa + i);
    const unsigned long long t0 = clock64();
    for (int it = 0; it < kSideIters; ++it) {
#pragma unroll
        for (int rep = 0; rep < kRepeat; ++rep)
#pragma unroll
            for (int i = 0; i < kChains; ++i)
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[i]) : "v"(aa), "v"(bb) : "vcc");
    }
    uint64_t r = 0;
    for (int i = 0; i < kChains; ++i) r ^= x[i];
    const unsigned long long t1 = clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
    if ((uint32_t)(r ^ (r >> 32)) == 0x12345678u) out[tid] = (uint32_t)r;
}

// --- LDS: the broadcast read used for the multiplier limbs, and ds_bpermute ---------------------
__global__ void __launch_bounds__(256) k_lds_bcast_b128(uint32_t* out, uint32_t seed) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[16 * 132];
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = threadIdx.x; i < 16 * 132; i += 256) lds[i] = seed + i;
    __syncthreads();
    const uint32_t* row = lds + (threadIdx.x >> 4) * 132;
    uint32_t r = 0;
    for (int it = 0; it < kIters / 4; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const uint4 v = *reinterpret_cast<const uint4*>(row + 4 * i);
            r += v.x ^ v.y ^ v.z ^ v.w;
            asm volatile("" : "+v"(r));
        }
    }
    if (r == 0x12345678u) out[tid] = r;
}
DEF_U32_TEST(k_ds_bpermute, "ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)")

// effective shader clock: s_memtime ticks (shader cycles) across a fixed VALU loop vs wall time
__global__ void __launch_bounds__(256) k_clock(unsigned long long* ticks, uint32_t seed) {
    uint32_t x = seed + threadIdx.x;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < kIters * 8; ++it) asm volatile("v_add_u32 %0, %0, %0" : "+v"(x));
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    if (x == 0x12345678u) ticks[0] = x;
}

struct Test {
    const char* name;
    void (*fn)(uint32_t*, uint32_t);
    double instr_per_iter;  // per lane per loop trip
};

int main() {
    int dev = 0;
    CK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * 8;  // 8 x 256 threads = 32 waves per CU
    uint32_t* d = nullptr;
    CK(hipMalloc((void**)&d, (size_t)blocks * 256 * 4));
    std::vector<Test> tests = {
        {"v_add_u32", k_add_u32, kChains},
        {"v_mov_b32", k_mov_b32, kChains},
        {"v_addc_co_u32", k_addc_chain, kChains},
        {"v_add_co+v_addc_pair", k_add_co_addc_pair, 2.0 * kChains},
        {"v_cndmask_b32", k_cndmask, kChains},
        {"v_cndmask_b32_e64_sgpr", k_cndmask_sgpr, kChains},
        {"v_and_b32", k_and_b32, kChains},
        {"v_alignbit_b32", k_alignbit, kChains},
        {"v_mad_u64_u32", k_mad_u64_u32, kChains},
        {"v_mad_u64_u32_banks_apart", k_mad_banks_apart, kChains},
        {"v_mad_u64_u32_banks_clash", k_mad_banks_clash, kChains},
        {"v_mad_u64_u32_srcs_same_bank", k_mad_srcs_same_bank, kChains},
        {"v_mad_u64_u32_banks_mixed", k_mad_banks_mixed, kChains},
        {"v_mul_lo_u32", k_mul_lo_u32, kChains},
        {"v_mul_hi_u32", k_mul_hi_u32, kChains},
        {"v_mad_u32_u24", k_mad_u32_u24, kChains},
        {"v_mul_hi_u32_u24", k_mul_hi_u32_u24, kChains},
        {"v_fma_f32", k_fma_f32, kChains},
        {"v_fmac_f32_vop2", k_fmac_f32, kChains},
        {"v_fma_f64", k_fma_f64, kChains},
        {"v_lshl_add_u64", k_lshl_add_u64, kChains},
        {"v_lshrrev_b64", k_lshrrev_b64, kChains},
        {"v_mov_b32_dpp_row_shl1", k_dpp_row_shl1, kChains},
        {"v_mov_b32_dpp_row_newbcast", k_dpp_newbcast, kChains},
        {"ds_bpermute_b32", k_ds_bpermute, kChains},
        {"ds_read_b128_row_broadcast", k_lds_bcast_b128, 32.0 / 4.0},
        // ops counted: MFMAs per lane-iteration (x kIters); 16x16x64 = 16384 int8 MACs per wave instruction
        {"mfma_i32_16x16x64_i8", k_mfma_i8, 1.0},
        // per trip: 8 v_mad_u64_u32 + 1 MFMA, counted as the 8 mads: compare with the v_mad_u64_u32 row
        {"mix_8mad+1mfma_i8(mads)", k_mix_mfma_mad, (double)kChains},
        {"mix_mad+addc", k_mix_mad_addc, 2.0 * kChains},
        {"mix_mad+addc+mov", k_mix_mad_addc_mov, 3.0 * kChains},
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // effective clock under an all-CU VALU load
    unsigned long long* dt = nullptr;
    CK(hipMalloc((void**)&dt, (size_t)blocks * 8));
    k_clock<<<blocks, 256>>>(dt, 1);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_clock<<<blocks, 256>>>(dt, 2);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float clk_ms = 0;
    CK(hipEventElapsedTime(&clk_ms, e0, e1));
    unsigned long long ticks0 = 0;
    CK(hipMemcpy(&ticks0, dt, 8, hipMemcpyDeviceToHost));
    // each wave issues kIters*8 dependent v_add; 8 waves per SIMD interleave, so the kernel lasts ~ticks0 cycles
    // cycles are also reported at the clock the chip really held, when the caller knows it (PHE_MB_CLOCK_MHZ: GRBM_GUI_ACTIVE
    // per XCD / kernel time from a rocprofv3 --pmc pass over this binary; tools/gpu_profile_round.sh does that)
    double measured_hz = prop.clockRate * 1e3;
    if (const char* e = getenv("PHE_MB_CLOCK_MHZ")) {
        const double mhz = atof(e);
        if (mhz > 500 && mhz < 4000) measured_hz = mhz * 1e6;
    }
    {   // one wave per SIMD issuing DEPENDENT full-rate VALU instructions: the time per instruction is the issue interval of a
        // lone wave (4 cycles if a wave64 instruction occupies its 16-lane SIMD for 4 cycles) — reported, not assumed
        CK(hipEventRecord(e0));
        k_clock<<<cus, 256>>>(dt, 3);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms1 = 0;
        CK(hipEventElapsedTime(&ms1, e0, e1));
        printf("{\"lone_wave_dependent_v_add_u32\": {\"kernel_ms\": %.4f, \"ns_per_instruction\": %.4f, \"cycles_at_clock_used\": %.3f}, "
               "\"clock_used_mhz\": %.1f, ", ms1, ms1 * 1e6 / ((double)kIters * 8), ms1 * 1e-3 * measured_hz / ((double)kIters * 8),
               measured_hz / 1e6);
    }
    printf("\"device\": \"%s\", \"gcn_arch\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, "
           "\"clock_probe\": {\"s_memtime_ticks\": %llu, \"kernel_ms\": %.4f, \"ticks_per_us\": %.1f}, \"tests\": {",
           prop.name, prop.gcnArchName, cus, prop.clockRate / 1000, ticks0, clk_ms, (double)ticks0 / (clk_ms * 1e3));
    bool first = true;
    for (const Test& t : tests) {
        t.fn<<<blocks, 256>>>(d, 1);  // warm-up
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            t.fn<<<blocks, 256>>>(d, rep + 2);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double lanes = (double)blocks * 256;
        const double wave_instr = lanes / 64.0 * kIters * t.instr_per_iter;
        const double lane_ops_per_s = lanes * kIters * t.instr_per_iter / (best * 1e-3);
        // cycles a SIMD spends per wave-instruction, at the nominal clock
        const double cyc = (double)cus * 4.0 * (prop.clockRate * 1e3) * (best * 1e-3) / wave_instr;
        printf("%s\"%s\": {\"ms\": %.4f, \"lane_ops_per_s\": %.4e, \"cycles_per_wave_instr_per_simd\": %.3f, "
               "\"cycles_at_measured_clock\": %.3f}",
               first ? "" : ", ", t.name, best, lane_ops_per_s, cyc, cyc * measured_hz / (prop.clockRate * 1e3));
        first = false;
    }
    printf("}, \"side_by_side_mfma_and_mad_waves\": {");
    {
        const int sblocks = cus;  // one 512-thread workgroup per CU: exactly two waves per SIMD
        bool f2 = true;
        for (int split = 0; split < 2; ++split) {
            for (int mode = 1; mode <= 3; ++mode) {
                k_side_by_side<<<sblocks, 512>>>(d, 1, mode, split);
                CK(hipDeviceSynchronize());
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipEventRecord(e0));
                    k_side_by_side<<<sblocks, 512>>>(d, rep + 2, mode, split);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms = 0;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                printf("%s\"%s_%s\": %.4f", f2 ? "" : ", ", split ? "odd_waves_matrix" : "waves_4to7_matrix",
                       mode == 1 ? "mad_only_ms" : mode == 2 ? "mfma_only_ms" : "both_ms", best);
                f2 = false;
            }
        }
        printf(", \"per_wave\": {\"mads\": %d, \"mfmas\": %d}", kSideIters * kRepeat * kChains, kSideIters * 16);
    }
    printf("}, \"side_by_side_saturated_valu\": {");
    {
        const double mads = (double)kSideIters * kRepeat * kChains, mfmas = (double)kSideIters * 16;
        for (int mode = 1; mode <= 3; ++mode) {
            CK(hipMemset(dt, 0, 16));
            k_side_by_side_saturated<<<cus, 768>>>(d, 1, mode, dt);
            CK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                k_side_by_side_saturated<<<cus, 768>>>(d, rep + 2, mode, dt);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            unsigned long long tk[2] = {0, 0};
            CK(hipMemcpy(tk, dt, 16, hipMemcpyDeviceToHost));
            printf("%s\"%s\": {\"ms\": %.4f, \"mad_wave_cycles_per_mad\": %.3f, \"matrix_wave_cycles_per_mfma\": %.3f}", mode == 1 ? "" : ", ",
                   mode == 1 ? "two_mad_waves_per_simd_alone" : mode == 2 ? "one_matrix_wave_per_simd_alone" : "both",
                   best, (mode & 1) ? (double)tk[0] / mads : 0.0, (mode & 2) ? (double)tk[1] / mfmas : 0.0);
        }
    }
    printf("}, \"mad_cycles_by_waves_per_simd\": {");
    for (int w = 1; w <= 4; w *= 2) {
        CK(hipMemset(dt, 0, 16));
        k_mad_cycles<<<cus, 256 * w>>>(d, 1, dt);
        CK(hipDeviceSynchronize());
        k_mad_cycles<<<cus, 256 * w>>>(d, 2, dt);
        CK(hipDeviceSynchronize());
        unsigned long long tk = 0;
        CK(hipMemcpy(&tk, dt, 8, hipMemcpyDeviceToHost));
        const double per_wave = (double)tk / ((double)kSideIters * kRepeat * kChains);
        printf("%s\"%d\": {\"cycles_per_mad_in_one_wave\": %.3f, \"simd_cycles_per_mad\": %.3f}", w == 1 ? "" : ", ", w, per_wave, per_wave / w);
    }
    printf("}}\n");
    CK(hipFree(d));
    return 0;
}
