// mix_probe.hip — what does an instruction that is NOT a multiply-add cost next to 8 multiply-adds?  (round 6)
//
// The headline sweeps issue, per row, 64 v_mad_u64_u32 and ~13 other vector instructions: DPP moves, 64-bit shifts and adds (all
// "half rate": ~4.2 cycles per wave64 instruction alone, profiles/microbench_r03c.json) and a few 32-bit logic operations ("full
// rate": ~2.4 cycles alone).  This probe times a "row" of 8 independent multiply-adds plus ONE candidate group of other instructions,
// at 1, 2 and 8 waves per SIMD, and prints the cycles the group ADDS to the row — i.e. whether full-rate 32-bit work hides in the
// shadow of the multiply-adds (then a half-rate 64-bit shift is worth replacing by three 32-bit operations) or not.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/mix_probe.hip -o /tmp/mix_probe && /tmp/mix_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CK(x)                                                       \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            return 1;                                               \
        }                                                           \
    } while (0)

constexpr int kTrips = 4096;  // loop trips; a trip = 8 rows of (8 multiply-adds + the extra group)

#define MADS                                                                                                   \
    "v_mad_u64_u32 %[x0], s[20:21], %[a], %[b], %[x0]\n\tv_mad_u64_u32 %[x1], s[20:21], %[a], %[b], %[x1]\n\t" \
    "v_mad_u64_u32 %[x2], s[20:21], %[a], %[b], %[x2]\n\tv_mad_u64_u32 %[x3], s[20:21], %[a], %[b], %[x3]\n\t" \
    "v_mad_u64_u32 %[x4], s[20:21], %[a], %[b], %[x4]\n\tv_mad_u64_u32 %[x5], s[20:21], %[a], %[b], %[x5]\n\t" \
    "v_mad_u64_u32 %[x6], s[20:21], %[a], %[b], %[x6]\n\tv_mad_u64_u32 %[x7], s[20:21], %[a], %[b], %[x7]\n\t"

// the same 8 multiply-adds with the carry-out (sdst, never read) on vcc / on 4 / on 8 different SGPR pairs
#define MADS_SD(S0, S1, S2, S3, S4, S5, S6, S7)                                                                \
    "v_mad_u64_u32 %[x0], " S0 ", %[a], %[b], %[x0]\n\tv_mad_u64_u32 %[x1], " S1 ", %[a], %[b], %[x1]\n\t" \
    "v_mad_u64_u32 %[x2], " S2 ", %[a], %[b], %[x2]\n\tv_mad_u64_u32 %[x3], " S3 ", %[a], %[b], %[x3]\n\t" \
    "v_mad_u64_u32 %[x4], " S4 ", %[a], %[b], %[x4]\n\tv_mad_u64_u32 %[x5], " S5 ", %[a], %[b], %[x5]\n\t" \
    "v_mad_u64_u32 %[x6], " S6 ", %[a], %[b], %[x6]\n\tv_mad_u64_u32 %[x7], " S7 ", %[a], %[b], %[x7]\n\t"
#define MADS_VCC MADS_SD("vcc", "vcc", "vcc", "vcc", "vcc", "vcc", "vcc", "vcc")
#define MADS_ROT4 MADS_SD("s[20:21]", "s[22:23]", "s[24:25]", "s[26:27]", "s[20:21]", "s[22:23]", "s[24:25]", "s[26:27]")
#define MADS_ROT8 MADS_SD("s[20:21]", "s[22:23]", "s[24:25]", "s[26:27]", "s[28:29]", "s[30:31]", "s[36:37]", "s[38:39]")
// multiplicands: 8 different register pairs instead of one (a, b) for all — %[p], %[q], %[t], %[m] join in
#define MADS_OPS                                                                                               \
    "v_mad_u64_u32 %[x0], s[20:21], %[a], %[b], %[x0]\n\tv_mad_u64_u32 %[x1], s[20:21], %[p], %[q], %[x1]\n\t" \
    "v_mad_u64_u32 %[x2], s[20:21], %[t], %[m], %[x2]\n\tv_mad_u64_u32 %[x3], s[20:21], %[a], %[q], %[x3]\n\t" \
    "v_mad_u64_u32 %[x4], s[20:21], %[p], %[b], %[x4]\n\tv_mad_u64_u32 %[x5], s[20:21], %[t], %[b], %[x5]\n\t" \
    "v_mad_u64_u32 %[x6], s[20:21], %[a], %[m], %[x6]\n\tv_mad_u64_u32 %[x7], s[20:21], %[p], %[m], %[x7]\n\t"
// multiplier from an SGPR (the whole-wave rungs and the tile kernels' fold)
#define MADS_SGPR                                                                                              \
    "v_mad_u64_u32 %[x0], s[20:21], s22, %[b], %[x0]\n\tv_mad_u64_u32 %[x1], s[20:21], s22, %[b], %[x1]\n\t" \
    "v_mad_u64_u32 %[x2], s[20:21], s22, %[b], %[x2]\n\tv_mad_u64_u32 %[x3], s[20:21], s22, %[b], %[x3]\n\t" \
    "v_mad_u64_u32 %[x4], s[20:21], s22, %[b], %[x4]\n\tv_mad_u64_u32 %[x5], s[20:21], s22, %[b], %[x5]\n\t" \
    "v_mad_u64_u32 %[x6], s[20:21], s22, %[b], %[x6]\n\tv_mad_u64_u32 %[x7], s[20:21], s22, %[b], %[x7]\n\t"

// y: a 64-bit register pair (y.lo = %L[y] is not available in clang inline asm: two 32-bit registers p, q and a pair y instead)
#define DEF_MIX(NAME, EXTRA) DEF_MIX2(NAME, MADS, EXTRA)
#define DEF_MIX2(NAME, MADS_, EXTRA)                                                                                  \
    __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {                                    \
        const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;                                                \
        uint32_t a = seed * 2654435761u + tid, b = (seed ^ tid) | 1u;                                              \
        uint64_t x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;        \
        uint64_t y = ((uint64_t)b << 32) | a, z = b;                                                               \
        uint32_t p = a ^ 5u, q = b ^ 9u, t = 0, m = 0x1fffffffu;                                                   \
        for (int it = 0; it < kTrips; ++it) {                                                                      \
            _Pragma("unroll") for (int rep = 0; rep < 8; ++rep)                                                    \
                asm volatile(MADS_ EXTRA                                                                           \
                             : [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3), [x4] "+v"(x4), [x5] "+v"(x5),   \
                               [x6] "+v"(x6), [x7] "+v"(x7), [y] "+v"(y), [z] "+v"(z), [p] "+v"(p), [q] "+v"(q), [t] "+v"(t) \
                             : [a] "v"(a), [b] "v"(b), [m] "v"(m)                                                  \
                             : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s36", "s37", "s38", "s39");                                                             \
        }                                                                                                          \
        const uint64_t r = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7 ^ y ^ z ^ p ^ q ^ t;                              \
        if ((uint32_t)(r ^ (r >> 32)) == 0x12345678u) out[tid] = (uint32_t)r;                                      \
    }

DEF_MIX(k_none, "")
DEF_MIX2(k_mads_vcc, MADS_VCC, "")
DEF_MIX2(k_mads_rot4, MADS_ROT4, "")
DEF_MIX2(k_mads_rot8, MADS_ROT8, "")
DEF_MIX2(k_mads_ops, MADS_OPS, "")
DEF_MIX2(k_mads_sgpr, MADS_SGPR, "")
DEF_MIX(k_shr64, "v_lshrrev_b64 %[y], 29, %[y]\n\t")
DEF_MIX(k_shr_3simple, "v_lshrrev_b32 %[t], 29, %[p]\n\tv_lshl_or_b32 %[p], %[q], 3, %[t]\n\tv_lshrrev_b32 %[q], 29, %[q]\n\t")
DEF_MIX(k_add64, "v_lshl_add_u64 %[y], %[y], 0, %[z]\n\t")
DEF_MIX(k_shr64_add64, "v_lshrrev_b64 %[y], 29, %[y]\n\tv_lshl_add_u64 %[z], %[z], 0, %[y]\n\t")
DEF_MIX(k_dpp_and, "v_and_b32_dpp %[p], %[q], %[m] quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t")
DEF_MIX(k_dpp_mov, "v_mov_b32_dpp %[p], %[q] quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t")
DEF_MIX(k_and1, "v_and_b32 %[p], %[p], %[m]\n\t")
DEF_MIX(k_and2, "v_and_b32 %[p], %[p], %[m]\n\tv_and_b32 %[q], %[q], %[m]\n\t")
DEF_MIX(k_and4, "v_and_b32 %[p], %[p], %[m]\n\tv_and_b32 %[q], %[q], %[m]\n\tv_and_b32 %[t], %[t], %[m]\n\tv_and_b32 %[p], %[p], %[a]\n\t")
DEF_MIX(k_and8, "v_and_b32 %[p], %[p], %[m]\n\tv_and_b32 %[q], %[q], %[m]\n\tv_and_b32 %[t], %[t], %[m]\n\tv_and_b32 %[p], %[p], %[a]\n\t"
                "v_and_b32 %[q], %[q], %[a]\n\tv_and_b32 %[t], %[t], %[a]\n\tv_and_b32 %[p], %[p], %[b]\n\tv_and_b32 %[q], %[q], %[b]\n\t")
DEF_MIX(k_lshl_or, "v_lshl_or_b32 %[p], %[q], 3, %[t]\n\t")
DEF_MIX(k_add3, "v_add3_u32 %[p], %[p], %[q], %[t]\n\t")
DEF_MIX(k_alignbit, "v_alignbit_b32 %[p], %[q], %[p], 29\n\t")
DEF_MIX(k_lshl32, "v_lshlrev_b32 %[p], 1, %[q]\n\t")
DEF_MIX(k_mul_lo, "v_mul_lo_u32 %[p], %[p], %[b]\n\t")
DEF_MIX(k_mad9, "v_mad_u64_u32 %[y], s[20:21], %[a], %[b], %[y]\n\t")
DEF_MIX(k_mov64, "v_mov_b64 %[y], %[z]\n\t")
DEF_MIX(k_add_co_pair, "v_add_co_u32 %[p], vcc, %[p], %[a]\n\tv_addc_co_u32 %[q], vcc, %[q], %[b], vcc\n\t")
// the non-multiply-add part of a real row of the fused squaring sweep: 4 DPP moves, 3 64-bit adds, 2 64-bit shifts, v_and, v_lshl
DEF_MIX(k_row_now, "v_and_b32_dpp %[p], %[q], %[m] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_and_b32 %[t], %[q], %[m]\n\tv_lshl_add_u64 %[z], %[z], 0, %[y]\n\t"
                   "v_and_b32_dpp %[q], %[p], %[m] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_and_b32_dpp %[t], %[q], %[m] quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_lshrrev_b64 %[y], 29, %[y]\n\tv_lshl_add_u64 %[z], %[z], 0, %[y]\n\t"
                   "v_and_b32_dpp %[p], %[t], %[m] quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_lshrrev_b64 %[y], 29, %[y]\n\tv_lshl_add_u64 %[z], %[z], 0, %[y]\n\tv_lshlrev_b32 %[t], 1, %[q]\n\t")

typedef void (*Kernel)(uint32_t*, uint32_t);
struct Test {
    const char* name;
    Kernel fn;
    int extra;  // instructions of the extra group
};

int main() {
    hipDeviceProp_t prop;
    CK(hipSetDevice(0));
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double hz = prop.clockRate * 1e3;
    uint32_t* d = nullptr;
    CK(hipMalloc((void**)&d, (size_t)cus * 8 * 256 * 4));
    const Test tests[] = {
        {"none", k_none, 0}, {"mads: sdst vcc", k_mads_vcc, -1}, {"mads: sdst 4 pairs", k_mads_rot4, -1}, {"mads: sdst 8 pairs", k_mads_rot8, -1},
        {"mads: 8 operand pairs", k_mads_ops, -1}, {"mads: SGPR multiplier", k_mads_sgpr, -1},
             {"shr64", k_shr64, 1},       {"shr_as_3_simple", k_shr_3simple, 3}, {"add64", k_add64, 1},
        {"shr64+add64", k_shr64_add64, 2}, {"and_dpp", k_dpp_and, 1}, {"mov_dpp", k_dpp_mov, 1},       {"and x1", k_and1, 1},
        {"and x2", k_and2, 2},        {"and x4", k_and4, 4},       {"and x8", k_and8, 8},              {"lshl_or", k_lshl_or, 1},
        {"add3", k_add3, 1},          {"alignbit", k_alignbit, 1}, {"lshl32", k_lshl32, 1},            {"mul_lo", k_mul_lo, 1},
        {"ninth mad", k_mad9, 1},     {"mov_b64", k_mov64, 1},     {"add_co+addc", k_add_co_pair, 2},  {"row as shipped (11)", k_row_now, 11},
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("cycles per row of 8 multiply-adds (+ the group), at the nominal clock %.0f MHz; added = minus the row of multiply-adds alone\n", hz / 1e6);
    for (int waves : {1, 2, 8}) {
        // `waves` waves per SIMD: workgroups of 256 threads put one wave on each SIMD of a CU
        const int blocks = cus * waves;
        double base = 0;
        printf("== %d wave(s) per SIMD\n", waves);
        for (const Test& t : tests) {
            t.fn<<<blocks, 256>>>(d, 1);
            CK(hipDeviceSynchronize());
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                t.fn<<<blocks, 256>>>(d, 2 + rep);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            const double rows = (double)kTrips * 8 * waves;  // rows a SIMD issues
            const double cyc = best * 1e-3 * hz / rows;
            if (t.extra == 0) base = cyc;
            printf("  %-22s %7.2f cycles per row   added %6.2f   (%d instruction%s: %5.2f each)\n", t.name, cyc, cyc - base, t.extra,
                   t.extra == 1 ? "" : "s", t.extra > 0 ? (cyc - base) / t.extra : 0.0);
        }
    }
    return 0;
}
