// phe_hip.hip — gfx950 kernels and the C-ABI (include/phe_hip.h) of the batched Paillier engine.
//
// Kernels (all hand-written for CDNA4; the arithmetic lives in mont_core.h / decrypt_tail.h):
//   k_modexp_uniform<G, L, MODE>  batch-uniform exponent: encrypt / obfuscate (e = n, mod n^2) and the
//                                 two CRT halves of decrypt (e = p-1 mod p^2, e = q-1 mod q^2)
//   k_modexp_split<G, L, MODE>    the same three jobs on the split-modulus pair representation (split_core.h):
//                                 half-width passes modulo n / p / q, ~1/3 fewer multiply-adds; the default engine
//                                 (PHE_HIP_ENGINE=full selects k_modexp_uniform instead)
//   k_modexp_var_split<G, L>      per-element exponent (powmod of _raw_mul) on the pair representation
//   k_modexp_var<G, L>            the same on the full-width modulus (PHE_HIP_ENGINE=full)
//   k_mulmod<G, L>                a*b mod n^2 (_raw_add, add-plaintext, the product tree of batched inversion)
//     (these three live in group_kernels.inc, instantiated by kernels_g*.hip, one TU per group width)
//   k_decrypt_tail                L-function, *hp, CRT recombination, one ciphertext per thread
//   k_select_rows                 per-row select between two ciphertext arrays
// Launch geometry: 256-thread workgroups = 4 wavefronts = 16 limb groups; the grid is sized to the
// resident capacity (CUs x blocks_per_cu) and each limb group strides over the batch, so the
// window tables (one per resident limb group) stay L2/MALL-resident regardless of batch size.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "../../include/phe_hip.h"
// clang-format off
#include "wave_gfx950.h"
#include "mont_core.h"
#include "split_core.h"
#include "mul_io.h"
#include "mul_table.h"
#include "decrypt_tail.h"
#include "key_setup.h"
#include "radix_conv.h"
#include "primality.h"
// clang-format on

#include <dlfcn.h>

using namespace phe;

struct phe_rccl_id {  // ncclUniqueId: 128 opaque bytes, passed by value (rccl.h NCCL_UNIQUE_ID_BYTES)
    char internal[128];
};
using host::Big;

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
constexpr int kBlock = 256;  // threads per workgroup = 4 wavefronts = 256/G limb groups (same constant as group_kernels.inc)

// The limb-group kernels live in kernels_g*.hip (compiled in parallel); every part answers -1 for an L it
// does not hold.
namespace phe {
#define PHE_DECLARE_PART(P)                                                                       \
    namespace P {                                                                                 \
    int occ_uniform(int L, int mode);                                                             \
    int launch_uniform(int L, int mode, int blocks, hipStream_t st, const UniformArgs& A);        \
    int occ_var(int L);                                                                           \
    int launch_var(int L, int blocks, hipStream_t st, const VarArgs& A);                          \
    int launch_mul(int L, int blocks, hipStream_t st, const MulArgs& A);                          \
    }
PHE_DECLARE_PART(g2a)
PHE_DECLARE_PART(g2b)
PHE_DECLARE_PART(g4a)
PHE_DECLARE_PART(g4b)
PHE_DECLARE_PART(g4c)
PHE_DECLARE_PART(g8a)
PHE_DECLARE_PART(g8b)
PHE_DECLARE_PART(g8c)
PHE_DECLARE_PART(g16a)
PHE_DECLARE_PART(g16b)
#undef PHE_DECLARE_PART
#define PHE_DECLARE_SPLIT_PART(P)                                                                 \
    namespace P {                                                                                 \
    int occ_split(int L, int mode);                                                               \
    int launch_split(int L, int mode, int blocks, hipStream_t st, const SplitArgs& A);            \
    int occ_var_split(int L);                                                                     \
    int launch_var_split(int L, int blocks, hipStream_t st, const SplitVarArgs& A);               \
    int occ_multi_split(int L);                                                                   \
    int launch_multi_split(int L, int blocks, hipStream_t st, const SplitMultiArgs& A);           \
    int launch_multi_tables(int L, int blocks, hipStream_t st, const SplitTableArgs& A);          \
    int launch_multi_lookup(int L, int blocks, hipStream_t st, const SplitLookupArgs& A);         \
    int launch_mul_split(int L, int blocks, hipStream_t st, const SplitMulArgs& A);               \
    int launch_crt_lift(int L, int blocks, hipStream_t st, const CrtLiftArgs& A);                 \
    int occ_split_unit(int L);                                                                    \
    int launch_split_unit(int L, int blocks, hipStream_t st, const SplitArgs& A);                 \
    int launch_pair(int L, int op, int blocks, hipStream_t st, const PairArgs& A);                \
    int launch_split_halves(int L, int blocks, hipStream_t st, const SplitArgs& Ap, const SplitArgs& Aq); \
    int launch_split_ab(int L, int mode, int numbers, int halves, hipStream_t st, const SplitArgs& Ap, const SplitArgs& Aq); \
    int launch_tail_wave(int L, int numbers, hipStream_t st, const TailWaveArgs& A);              \
    int occ_split_late(int L, int mode);                                                          \
    int launch_split_late(int L, int mode, int blocks, hipStream_t st, const SplitArgs& A);       \
    int launch_split_late_halves(int L, int blocks, hipStream_t st, const SplitArgs& Ap, const SplitArgs& Aq); \
    int occ_split_quick(int L, int mode);                                                         \
    int launch_split_quick(int L, int mode, int blocks, hipStream_t st, const SplitArgs& A);      \
    int launch_split_quick_halves(int L, int blocks, hipStream_t st, const SplitArgs& Ap, const SplitArgs& Aq); \
    }
PHE_DECLARE_SPLIT_PART(s1a)
PHE_DECLARE_SPLIT_PART(s2a)
PHE_DECLARE_SPLIT_PART(s2b)
PHE_DECLARE_SPLIT_PART(s2c)
PHE_DECLARE_SPLIT_PART(s4a)
PHE_DECLARE_SPLIT_PART(s4b)
PHE_DECLARE_SPLIT_PART(s4c)
PHE_DECLARE_SPLIT_PART(s4d)
PHE_DECLARE_SPLIT_PART(s8a)
PHE_DECLARE_SPLIT_PART(s8b)
PHE_DECLARE_SPLIT_PART(s8c)
PHE_DECLARE_SPLIT_PART(s8d)
PHE_DECLARE_SPLIT_PART(s16a)
PHE_DECLARE_SPLIT_PART(s16b)
PHE_DECLARE_SPLIT_PART(s16c)
PHE_DECLARE_SPLIT_PART(s64a)
PHE_DECLARE_SPLIT_PART(s64b)
#undef PHE_DECLARE_SPLIT_PART
}  // namespace phe

namespace phe {
namespace t16 {  // kernels_t16.hip
int launch_mul_table(int L, int blocks, size_t lds_bytes, hipStream_t st, const TableMulArgs& A);
int launch_mul_tile(int L, int blocks, hipStream_t st, const TableMulArgs& A);
int tile_blocks_per_cu(int waves);
}  // namespace t16
}  // namespace phe

namespace phe {
namespace radix {  // kernels_radix.hip
size_t tile_bytes(int words);
int launch_to_decimal(const uint32_t* limbs, int words, char* digits, int width, uint64_t batch, unsigned long long* bad,
                      int max_blocks, hipStream_t st);
int launch_from_decimal(const char* digits, int width, uint32_t* limbs, int words, uint64_t batch,
                        unsigned long long* bad_char, unsigned long long* bad_size, int max_blocks, hipStream_t st);
}  // namespace radix
}  // namespace phe

namespace phe {
namespace mr {  // kernels_mr.hip
int launch(int L, int blocks, hipStream_t st, const MillerRabinArgs& A);
}  // namespace mr
}  // namespace phe

struct KernelPart {
    int G;
    int (*occ_uniform)(int, int);
    int (*launch_uniform)(int, int, int, hipStream_t, const UniformArgs&);
    int (*occ_var)(int);
    int (*launch_var)(int, int, hipStream_t, const VarArgs&);
    int (*launch_mul)(int, int, hipStream_t, const MulArgs&);
};
static const KernelPart kParts[] = {
    {2, phe::g2a::occ_uniform, phe::g2a::launch_uniform, phe::g2a::occ_var, phe::g2a::launch_var, phe::g2a::launch_mul},
    {2, phe::g2b::occ_uniform, phe::g2b::launch_uniform, phe::g2b::occ_var, phe::g2b::launch_var, phe::g2b::launch_mul},
    {4, phe::g4a::occ_uniform, phe::g4a::launch_uniform, phe::g4a::occ_var, phe::g4a::launch_var, phe::g4a::launch_mul},
    {4, phe::g4b::occ_uniform, phe::g4b::launch_uniform, phe::g4b::occ_var, phe::g4b::launch_var, phe::g4b::launch_mul},
    {4, phe::g4c::occ_uniform, phe::g4c::launch_uniform, phe::g4c::occ_var, phe::g4c::launch_var, phe::g4c::launch_mul},
    {8, phe::g8a::occ_uniform, phe::g8a::launch_uniform, phe::g8a::occ_var, phe::g8a::launch_var, phe::g8a::launch_mul},
    {8, phe::g8b::occ_uniform, phe::g8b::launch_uniform, phe::g8b::occ_var, phe::g8b::launch_var, phe::g8b::launch_mul},
    {8, phe::g8c::occ_uniform, phe::g8c::launch_uniform, phe::g8c::occ_var, phe::g8c::launch_var, phe::g8c::launch_mul},
    {16, phe::g16a::occ_uniform, phe::g16a::launch_uniform, phe::g16a::occ_var, phe::g16a::launch_var, phe::g16a::launch_mul},
    {16, phe::g16b::occ_uniform, phe::g16b::launch_uniform, phe::g16b::occ_var, phe::g16b::launch_var, phe::g16b::launch_mul},
};

struct SplitPart {
    int G;
    int (*occ_split)(int, int);
    int (*launch_split)(int, int, int, hipStream_t, const SplitArgs&);
    int (*occ_var_split)(int);
    int (*launch_var_split)(int, int, hipStream_t, const SplitVarArgs&);
    int (*occ_multi_split)(int);
    int (*launch_multi_split)(int, int, hipStream_t, const SplitMultiArgs&);
    int (*launch_multi_tables)(int, int, hipStream_t, const SplitTableArgs&);
    int (*launch_multi_lookup)(int, int, hipStream_t, const SplitLookupArgs&);
    int (*launch_mul_split)(int, int, hipStream_t, const SplitMulArgs&);
    int (*launch_crt_lift)(int, int, hipStream_t, const CrtLiftArgs&);
    int (*occ_split_unit)(int);
    int (*launch_split_unit)(int, int, hipStream_t, const SplitArgs&);
    int (*launch_pair)(int, int, int, hipStream_t, const PairArgs&);
    int (*launch_split_halves)(int, int, hipStream_t, const SplitArgs&, const SplitArgs&);
    int (*launch_split_ab)(int, int, int, int, hipStream_t, const SplitArgs&, const SplitArgs&);
    int (*launch_tail_wave)(int, int, hipStream_t, const TailWaveArgs&);
    int (*occ_split_late)(int, int);
    int (*launch_split_late)(int, int, int, hipStream_t, const SplitArgs&);
    int (*launch_split_late_halves)(int, int, hipStream_t, const SplitArgs&, const SplitArgs&);
    int (*occ_split_quick)(int, int);
    int (*launch_split_quick)(int, int, int, hipStream_t, const SplitArgs&);
    int (*launch_split_quick_halves)(int, int, hipStream_t, const SplitArgs&, const SplitArgs&);
};
#define PHE_SPLIT_PART(NS, G_)                                                                                                   \
    {G_, phe::NS::occ_split, phe::NS::launch_split, phe::NS::occ_var_split, phe::NS::launch_var_split, phe::NS::occ_multi_split,  \
     phe::NS::launch_multi_split, phe::NS::launch_multi_tables, phe::NS::launch_multi_lookup, phe::NS::launch_mul_split,        \
     phe::NS::launch_crt_lift, phe::NS::occ_split_unit, phe::NS::launch_split_unit, phe::NS::launch_pair,                       \
     phe::NS::launch_split_halves, phe::NS::launch_split_ab, phe::NS::launch_tail_wave, phe::NS::occ_split_late,               \
     phe::NS::launch_split_late, phe::NS::launch_split_late_halves, phe::NS::occ_split_quick, phe::NS::launch_split_quick,      \
     phe::NS::launch_split_quick_halves}
static const SplitPart kSplitParts[] = {
    PHE_SPLIT_PART(s1a, 1),
    PHE_SPLIT_PART(s2a, 2),
    PHE_SPLIT_PART(s2b, 2),
    PHE_SPLIT_PART(s2c, 2),
    PHE_SPLIT_PART(s4a, 4),
    PHE_SPLIT_PART(s4b, 4),
    PHE_SPLIT_PART(s4c, 4),
    PHE_SPLIT_PART(s4d, 4),
    PHE_SPLIT_PART(s8a, 8),
    PHE_SPLIT_PART(s8b, 8),
    PHE_SPLIT_PART(s8c, 8),
    PHE_SPLIT_PART(s8d, 8),
    PHE_SPLIT_PART(s16a, 16),
    PHE_SPLIT_PART(s16b, 16),
    PHE_SPLIT_PART(s16c, 16),
    PHE_SPLIT_PART(s64a, 64),
    PHE_SPLIT_PART(s64b, 64),
};
#undef PHE_SPLIT_PART
#define PHE_SPLIT_BY_GROUP(G_, CALL2)                 \
    [&]() -> int {                                    \
        for (const SplitPart& part_ : kSplitParts) {  \
            if (part_.G != (G_)) continue;            \
            const int r_ = part_.CALL2;               \
            if (r_ >= 0) return r_;                   \
        }                                             \
        return -1;                                    \
    }()

// first part of group width G_ that holds the requested L (its call returns >= 0); -1 if none does
#define PHE_BY_GROUP(G_, CALL2)                       \
    [&]() -> int {                                    \
        for (const KernelPart& part_ : kParts) {      \
            if (part_.G != (G_)) continue;            \
            const int r_ = part_.CALL2;               \
            if (r_ >= 0) return r_;                   \
        }                                             \
        return -1;                                    \
    }()

constexpr int kTailBlock = 64;
// threads per workgroup of the CRT tail: one wavefront, or half of one where p, q are so wide (4096 bits: 8192-bit keys)
// that 64 per-thread workspaces exceed the LDS of a CU
static int tail_block(int h) { return ((size_t)tail_ws_words(h) * kTailBlock * 4 > 150 * 1024) ? kTailBlock / 2 : kTailBlock; }
__global__ void __launch_bounds__(kTailBlock) k_decrypt_tail(TailArgs A) {
    extern __shared__ __attribute__((aligned(16))) uint32_t tail_ws[];
    const uint64_t item = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= A.batch) return;
    const TailWs ws{tail_ws + threadIdx.x, (int)blockDim.x};
    decrypt_tail_one(A, ws, item);
}

// out[i] = mask[i] ? b[i] : a[i], rows of `limbs` 32-bit words (the branch select of _raw_mul, phe/paillier.py:745-751)
__global__ void __launch_bounds__(256) k_select_rows(const uint32_t* a, const uint32_t* b, const uint8_t* mask,
                                                     uint32_t* out, int limbs, uint64_t batch) {
    const uint64_t total = batch * (uint64_t)limbs;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t row = i / (uint64_t)limbs;
        out[i] = mask[row] ? b[i] : a[i];
    }
}

// rows of `limbs` words moved by an index list: gather dst[j] = src[idx[j]], scatter dst[idx[j]] = src[j] (j < count) — the
// negative-scalar branch of _raw_mul inverts only the rows that take it (phe/paillier.py:745-749)
// far_rows: rows of the indexed side (gather: src, scatter: dst).  An index beyond it touches nothing on that side: the scatter skips
// the row, the gather fills it with zeros (never a read or write outside the caller's buffer for a bad index of the public API).
template <bool SCATTER>
__global__ void __launch_bounds__(256) k_move_rows(const uint32_t* src, const uint32_t* idx, uint32_t* dst, int limbs, uint64_t count,
                                                   uint64_t far_rows) {
    const uint64_t total = count * (uint64_t)limbs;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t j = i / (uint64_t)limbs, w = i - j * (uint64_t)limbs;
        const uint64_t row = idx[j];
        const bool inside = row < far_rows;
        const uint64_t far = row * (uint64_t)limbs + w;
        if (SCATTER) {
            if (inside) dst[far] = src[i];
        } else {
            dst[i] = inside ? src[far] : 0u;
        }
    }
}

__global__ void k_selftest_prims(uint32_t* out) {
    const uint32_t lane = threadIdx.x & 63u;
    const wave::Lanes<16> l16(lane);
    const wave::Lanes<8> l8(lane);
    out[lane] = wave::grp_down1<16>(lane + 100u, l16);
    out[64 + lane] = wave::grp_up1<16>(lane + 100u, l16);
    out[128 + lane] = wave::grp_bcast0<16>(lane + 100u, l16);
    const uint64_t b = wave::ballot((lane % 3u) == 0u);
    out[192 + lane] = (uint32_t)(b >> lane) & 1u;
    if (lane < 2) out[256 + lane] = (uint32_t)(b >> (32 * lane));
    out[258 + lane] = wave::grp_down1<8>(lane + 100u, l8);
    out[322 + lane] = wave::grp_up1<8>(lane + 100u, l8);
    out[386 + lane] = wave::grp_bcast0<8>(lane + 100u, l8);
    const wave::Lanes<4> l4(lane);
    out[450 + lane] = wave::grp_down1<4>(lane + 100u, l4);
    out[514 + lane] = wave::grp_up1<4>(lane + 100u, l4);
    out[578 + lane] = wave::grp_bcast0<4>(lane + 100u, l4);
    const wave::Lanes<2> l2(lane);
    out[642 + lane] = wave::grp_down1<2>(lane + 100u, l2);
    out[706 + lane] = wave::grp_up1<2>(lane + 100u, l2);
    out[770 + lane] = wave::grp_bcast0<2>(lane + 100u, l2);
    const wave::Lanes<64> l64(lane);  // the whole wavefront as one group: wave_shl / wave_shr / readfirstlane
    out[834 + lane] = wave::grp_down1<64>(lane + 100u, l64);
    out[898 + lane] = wave::grp_up1<64>(lane + 100u, l64);
    out[962 + lane] = wave::grp_bcast0<64>(lane + 100u, l64);
}

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
// what phe_hip_ctx_last_launch reports (include/phe_hip.h)
enum : int { kPathUnit = 1, kPathOwner = 2, kPathSideBySide = 4, kPathPipelined = 8, kPathFusedObfuscate = 16, kPathWavePairs = 32, kPathWaveTail = 64,
             kPathLate = 128, kPathTableMul = 256, kPathTileMul = 512 };
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(PHE_HIP_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));    \
    } while (0)

struct DevModulus {  // device copies of a host::ModulusPack
    int G = 0, L = 0, S = 0;
    uint32_t* blob = nullptr;  // n | r1 | r2 | r3 | aux
    ModConsts c{};
};
struct DevSplit {  // device copies of a host::SplitPack
    int G = 0, L = 0, H = 0;  // G == 0: no split kernel for this modulus
    int rows = 0;             // limbs of a number (H, except on the whole-wave geometry: SplitPack::rows)
    uint32_t* blob = nullptr;  // n | r1 | e | nsq | conv [| the QuickPack's rows]
    SplitConsts c{};
    // one number on a wave pair (host::QuickPack; whole-wave packs only): constants modulo the scaled modulus and, for the
    // way out, modulo the true one; q_L == 0: not offered
    int q_L = 0, q_H = 0, q_rows = 0;
    SplitConsts q_scaled{}, q_exit{};
};
struct DevSchedule {
    uint32_t* ops = nullptr;
    int n_ops = 0, first_idx = 0, tbl_entries = 1;
};
struct DevTail {
    uint32_t* blob = nullptr;
    TailConsts k{};
};

struct phe_hip_ctx {
    int device = 0;
    int n_cus = 256;
    std::string arch;  // hipDeviceProp_t::gcnArchName ("gfx950:sramecc+:xnack-"): a measured ladder names the arch it was made on
    int blocks_per_cu = 0;  // 0 = ask the occupancy API per kernel (PHE_HIP_BLOCKS_PER_CU / set_blocks_per_cu override)
    int prefer_group = 0;  // 0 = automatic geometry; PHE_HIP_GROUP=4|8|16 sets the narrowest limb group allowed
    bool has_private = false;
    host::PublicPlan pub;
    host::PrivatePlan priv;
    DevModulus d_nsq, d_psq, d_qsq;
    bool use_split = true;  // PHE_HIP_ENGINE=full: keep the uniform-exponent jobs on the full-width kernels
    DevSplit d_nsplit, d_psplit, d_qsplit;
    DevSplit d_nunit;                 // the scaled modulus n' = k*n (key_setup.h PublicPlan::nunit); G == 0: not offered
    uint32_t* unit_tmp = nullptr;     // r^n mod n'^2, rows of pub.unit_words words
    size_t unit_tmp_words = 0;
    // Geometry ladder.  The members above (pub / d_nsq / d_nsplit, priv / d_psq ...) are rung 0: the narrowest limb groups,
    // i.e. the most limbs per lane = the fewest instructions per multiply-add = the best throughput once every SIMD has a
    // wave.  A batch too small for that (batch * G lanes < the chip's lanes) climbs to wider groups — fewer limbs per lane,
    // shorter dependent chains per product, more lanes per number — until the chip is filled or the groups are 16 wide
    // (pick_rung).  The constants of a modulus are rows of G*L limbs in global limb order, so every rung has its own copies.
    struct PubRung {
        host::PublicPlan plan;
        DevModulus nsq;
        DevSplit nsplit;
        DevSplit nunit;  // the scaled modulus on this rung's geometry (G == 0: not offered there)
    };
    struct PrivRung {
        host::PrivatePlan plan;
        DevModulus psq, qsq;
        DevSplit psplit, qsplit;
    };
    std::vector<PubRung> pub_rungs;    // rungs 1.. of the public side, by increasing group width
    std::vector<PrivRung> priv_rungs;  // rungs 1.. of the private side
    int wave_pair_depth = 1;           // waves per SIMD up to which a batch stays on the wave-pair kernels (PHE_HIP_WAVE_PAIR_DEPTH)
    bool no_wave_pairs = false;        // PHE_HIP_NO_WAVE_PAIRS=1: a handful of numbers stays on the single-wave kernels (A/B measurements, tests)
    // a*b mod n^2 as one plain product + one fold against a table in LDS (csrc/mul_table.h): constants n | W^S - n^2 | its shift |
    // table on the device; null = not offered for this key width (the table does not fit a CU's LDS) or PHE_HIP_NO_TABLE_MUL=1
    host::TableMulPack tmul;
    uint32_t* tmul_blob = nullptr;
    uint32_t* tmul_cols = nullptr;  // the table in mul_tile.h's column-block layout (nullptr: that kernel is not offered)
    bool no_late = false;              // PHE_HIP_NO_LATE=1: the small-batch rungs stay on the round-3 kernels (textbook row order; A/B measurements, tests)
    bool force_unit = false;           // PHE_HIP_FORCE_UNIT=1: r^n through the scaled modulus whatever the batch size (tests)
    int force_group = 0;               // phe_hip_ctx_set_group: 0 = pick by batch size; G = the rung whose groups are G lanes wide
    // the measured ladder (phe_hip_ctx_load_ladder): per (kernel family, group width) the launch time at a few batch sizes, made by
    // `tools/bench_sweep.py --calibrate` with every rung pinned.  Key: family * 256 + G.  Empty: the estimate of rung_cost.
    struct LadderPoint { double rows, ns; };
    std::map<int, std::vector<LadderPoint>> ladder;
    // what the last launch on this context took (phe_hip_ctx_last_launch): tests assert the path they meant to exercise
    int last_path = 0, last_geom_pub = 0, last_geom_priv = 0;
    // one stream order per context: every *_dev call waits for the previous call's work when it is issued on another
    // stream (the window tables and intermediates are the context's, not the call's)
    hipEvent_t ev_busy = nullptr;
    hipStream_t busy_stream = nullptr;
    bool busy_valid = false;
    DevSchedule d_exp_n, d_exp_p, d_exp_q;
    DevTail d_tail;
    // the CRT tail on one wavefront per ciphertext (small batches; key_setup.h TailWavePack); tail_wave_L == 0: not offered
    host::TailWavePack tail_wave;
    uint32_t* tail_wave_blob = nullptr;
    TailWaveConsts d_tail_wave{};
    int tail_wave_L = 0;
    size_t tail_wave_per_cu = 16;  // batches of up to this many ciphertexts per CU take it (create_private; PHE_HIP_WAVE_TAIL_PER_CU)
    // grow-only device scratch
    uint32_t* table = nullptr;
    size_t table_words = 0;
    uint32_t* scratch = nullptr;  // decrypt intermediates x_p | x_q
    size_t scratch_words = 0;
    unsigned long long* flags = nullptr;  // radix conversion: first offending row per error kind (2 words)
    uint32_t* table2 = nullptr;  // window tables of the q half when the two halves of a small decrypt run concurrently
    size_t table2_words = 0;
    uint32_t* lookup = nullptr;  // multi-exponentiation: the 2^w-ary tables of a whole vector (phe_hip_multiexp_csr_dev)
    size_t lookup_words = 0;
    uint32_t* item_sched = nullptr;  // per-number exponent schedules of a small phe_hip_powmod (ops of all numbers | 4 words of meta each)
    size_t item_sched_words = 0;
    uint32_t* partial = nullptr;  // multi-exponentiation: one product per chunk, joined in place by a k_mulmod tree
    size_t partial_words = 0;
    // encryption by the key owner (CRT lift): K*R, (q^2 - K)*R mod q^2 and p^2 as rows of d_qsq.S limbs; null = not offered
    uint32_t* owner_blob = nullptr;
    int owner_G = 0, owner_L = 0;
    // staging for the host-pointer entry points
    uint32_t* stage[3] = {nullptr, nullptr, nullptr};
    size_t stage_words[3] = {0, 0, 0};
    // ... and for a handful of rows: one pinned, device-mapped buffer of three slots that the kernels read and write across
    // PCIe themselves (stage_in / stage_out below); cur[] = where the current call's operands are, either kind
    uint32_t* mapped_host = nullptr;
    uint32_t* mapped_dev = nullptr;
    bool no_mapped = false;
    uint32_t* cur[3] = {nullptr, nullptr, nullptr};
    // large host batches: chunks double-buffered through pinned memory, uploads / kernels / downloads on three streams
    struct HostPipe {
        hipStream_t s_in = nullptr, s_comp = nullptr, s_out = nullptr;
        hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_comp[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
        uint32_t* pin[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};   // pinned host: in0 | in1 | out
        size_t pin_words[2][3] = {{0, 0, 0}, {0, 0, 0}};
        uint32_t* dev[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};   // device:      in0 | in1 | out
        size_t dev_words[2][3] = {{0, 0, 0}, {0, 0, 0}};
        bool ready = false;
    } pipe;
};

static int upload_modulus(const host::ModulusPack& m, DevModulus& d) {
    d.G = m.G;
    d.L = m.L;
    d.S = m.S;
    if (m.G == 0) return PHE_HIP_OK;  // no full-width geometry for this modulus (wide keys: the pair form serves it)
    const size_t words = (size_t)5 * m.S;
    HIP_TRY(hipMalloc((void**)&d.blob, words * 4));
    std::vector<uint32_t> h(words);
    const std::vector<uint32_t>* parts[5] = {&m.n, &m.r1, &m.r2, &m.r3, &m.aux};
    for (int i = 0; i < 5; ++i) memcpy(h.data() + (size_t)i * m.S, parts[i]->data(), (size_t)m.S * 4);
    HIP_TRY(hipMemcpy(d.blob, h.data(), words * 4, hipMemcpyHostToDevice));
    d.c.n = d.blob;
    d.c.r1 = d.blob + m.S;
    d.c.r2 = d.blob + 2 * m.S;
    d.c.r3 = d.blob + 3 * m.S;
    d.c.aux = d.blob + 4 * m.S;
    d.c.n0inv = m.n0inv;
    return PHE_HIP_OK;
}
static int upload_split(const host::SplitPack& m, DevSplit& d, const host::QuickPack* quick = nullptr) {
    d.G = m.G;
    d.L = m.L;
    d.H = m.H;
    d.rows = m.rows;
    if (m.G == 0) return PHE_HIP_OK;
    std::vector<uint32_t> h;
    const bool q = quick && quick->ok() && m.G >= 16 && quick->scaled.G == m.G;
    const std::vector<uint32_t>* parts[15] = {&m.n, &m.r1, &m.e, &m.nsq, &m.conv};
    int n_parts = 5;
    if (q) {
        const host::SplitPack &S = quick->scaled, &E = quick->exit;
        const std::vector<uint32_t>* more[10] = {&S.n, &S.r1, &S.e, &S.nsq, &S.conv, &quick->nbar, &E.n, &E.r1, &E.nsq, &quick->kx};
        for (int i = 0; i < 10; ++i) parts[n_parts++] = more[i];
    }
    size_t off[15];
    for (int i = 0; i < n_parts; ++i) {
        off[i] = h.size();
        h.insert(h.end(), parts[i]->begin(), parts[i]->end());
    }
    HIP_TRY(hipMalloc((void**)&d.blob, h.size() * 4));
    HIP_TRY(hipMemcpy(d.blob, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    d.c.n = d.blob + off[0];
    d.c.r1 = d.blob + off[1];
    d.c.e = d.blob + off[2];
    d.c.nsq = d.blob + off[3];
    d.c.conv = d.blob + off[4];
    d.c.n0inv = m.n0inv;
    d.c.rows = m.rows;
    if (q) {
        const host::SplitPack &S = quick->scaled, &E = quick->exit;
        d.q_L = S.L;
        d.q_H = S.H;
        d.q_rows = S.rows;
        d.q_scaled.n = d.blob + off[5];
        d.q_scaled.r1 = d.blob + off[6];
        d.q_scaled.e = d.blob + off[7];
        d.q_scaled.nsq = d.blob + off[8];
        d.q_scaled.conv = d.blob + off[9];
        d.q_scaled.nbar = d.blob + off[10];
        d.q_scaled.n0inv = S.n0inv;
        d.q_scaled.rows = S.rows;
        d.q_exit.n = d.blob + off[11];
        d.q_exit.r1 = d.blob + off[12];
        d.q_exit.nsq = d.blob + off[13];
        d.q_exit.kx = d.blob + off[14];
        d.q_exit.e = d.q_exit.conv = nullptr;
        d.q_exit.n0inv = E.n0inv;
        d.q_exit.rows = E.rows;
    }
    return PHE_HIP_OK;
}
static int upload_schedule(const host::Schedule& s, DevSchedule& d) {
    d.n_ops = (int)s.ops.size();
    d.first_idx = s.first_idx;
    d.tbl_entries = s.tbl_entries;
    const size_t bytes = std::max<size_t>(1, s.ops.size()) * 4;
    HIP_TRY(hipMalloc((void**)&d.ops, bytes));
    if (!s.ops.empty()) HIP_TRY(hipMemcpy(d.ops, s.ops.data(), s.ops.size() * 4, hipMemcpyHostToDevice));
    return PHE_HIP_OK;
}
static int upload_tail(const host::TailPack& t, DevTail& d) {
    const int h = t.h;
    std::vector<uint32_t> hbuf((size_t)7 * h);
    const Big* parts[7] = {&t.p, &t.q, &t.pinvw, &t.qinvw, &t.hp_r, &t.hq_r, &t.pinvq_r};
    for (int i = 0; i < 7; ++i) memcpy(hbuf.data() + (size_t)i * h, parts[i]->data(), (size_t)h * 4);
    HIP_TRY(hipMalloc((void**)&d.blob, hbuf.size() * 4));
    HIP_TRY(hipMemcpy(d.blob, hbuf.data(), hbuf.size() * 4, hipMemcpyHostToDevice));
    d.k.h = h;
    d.k.p = d.blob;
    d.k.q = d.blob + h;
    d.k.pinvw = d.blob + 2 * h;
    d.k.qinvw = d.blob + 3 * h;
    d.k.hp_r = d.blob + 4 * h;
    d.k.hq_r = d.blob + 5 * h;
    d.k.pinvq_r = d.blob + 6 * h;
    d.k.p0inv = t.p0inv;
    d.k.q0inv = t.q0inv;
    return PHE_HIP_OK;
}

static int upload_tail_wave(phe_hip_ctx* ctx) {
    ctx->tail_wave = host::build_tail_wave(ctx->priv.tail);
    const host::TailWavePack& W = ctx->tail_wave;
    if (!W.ok()) return PHE_HIP_OK;
    const std::vector<uint32_t>* parts[7] = {&W.p, &W.q, &W.pinv, &W.qinv, &W.hp_r, &W.hq_r, &W.pinvq_r};
    std::vector<uint32_t> h;
    for (int i = 0; i < 7; ++i) h.insert(h.end(), parts[i]->begin(), parts[i]->end());
    HIP_TRY(hipMalloc((void**)&ctx->tail_wave_blob, h.size() * 4));
    HIP_TRY(hipMemcpy(ctx->tail_wave_blob, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const uint32_t* b = ctx->tail_wave_blob;
    TailWaveConsts& k = ctx->d_tail_wave;
    k.p = b;
    k.q = b + W.H;
    k.pinv = b + 2 * W.H;
    k.qinv = b + 3 * W.H;
    k.hp_r = b + 4 * W.H;
    k.hq_r = b + 5 * W.H;
    k.pinvq_r = b + 6 * W.H;
    k.p0inv = W.p0inv;
    k.q0inv = W.q0inv;
    k.rows = W.rows;
    ctx->tail_wave_L = W.L;
    return PHE_HIP_OK;
}

static int ensure_words(uint32_t** buf, size_t* have, size_t need) {
    if (need <= *have) return PHE_HIP_OK;
    if (*buf) HIP_TRY(hipFree(*buf));
    *buf = nullptr;
    *have = 0;
    HIP_TRY(hipMalloc((void**)buf, need * 4));
    *have = need;
    return PHE_HIP_OK;
}

static int grid_blocks(const phe_hip_ctx* ctx, size_t batch, int G, int blocks_per_cu) {
    const size_t groups = (size_t)(kBlock / G);
    const size_t want = (batch + groups - 1) / groups;
    const size_t cap = (size_t)ctx->n_cus * (size_t)std::max(1, blocks_per_cu);
    return (int)std::max<size_t>(1, std::min(want, cap));
}

// ------------------------------------------------------------------------------------------------
// launches (device pointers)
// ------------------------------------------------------------------------------------------------
// The constants of a modulus are rows of S = G*L limbs in global limb order, so two geometries with the same S
// can share them.  The 4x36 split only pays in the uniform-exponent kernel (whose loops fit 256 VGPRs); the
// per-element-exponent and mulmod kernels keep more operands alive and run the same numbers as 8x18.
static void light_geometry(const DevModulus& M, int& G, int& L) {
    G = M.G;
    L = M.L;
    if (G == 4 && L == 36) {         // 144 limbs
        G = 8;
        L = 18;
    } else if (G == 2 && L == 36) {  // 72 limbs
        G = 4;
        L = 18;
    }
}

// Resident workgroups per CU for a kernel (VGPR/LDS-limited), asked once per instantiation.  The modexp
// kernels size their grid to the residency so that the window tables stay per resident group.
template <int MODE>
static int launch_uniform(phe_hip_ctx* ctx, const DevModulus& M, const DevSchedule& E, const uint32_t* base,
                          int base_limbs, const uint32_t* post, int post_limbs, uint32_t* out, int out_limbs,
                          size_t batch, hipStream_t stream) {
    int per_cu = ctx->blocks_per_cu;
    if (per_cu == 0) per_cu = PHE_BY_GROUP(M.G, occ_uniform(M.L, MODE));
    if (per_cu < 0) return fail(PHE_HIP_EINVAL, "unsupported limb-group geometry");
    const int blocks = grid_blocks(ctx, batch, M.G, per_cu);
    const size_t rows = (size_t)blocks * (size_t)(kBlock / M.G);
    int rc = ensure_words(&ctx->table, &ctx->table_words, rows * (size_t)E.tbl_entries * M.S);
    if (rc) return rc;
    UniformArgs A;
    A.mod = M.c;
    A.sched = E.ops;
    A.n_ops = E.n_ops;
    A.first_idx = E.first_idx;
    A.tbl_entries = E.tbl_entries;
    A.base = base;
    A.base_limbs = base_limbs;
    A.post = post;
    A.post_limbs = post_limbs;
    A.out = out;
    A.out_limbs = out_limbs;
    A.table = ctx->table;
    A.batch = batch;
    if (PHE_BY_GROUP(M.G, launch_uniform(M.L, MODE, blocks, stream, A)) < 0)
        return fail(PHE_HIP_EINVAL, "unsupported limb-group geometry");
    HIP_TRY(hipGetLastError());
    return PHE_HIP_OK;
}

static int chunks_for(int limbs32, int H) { return std::max(1, (32 * limbs32 + 29 * H - 1) / (29 * H)); }

// arguments of one k_modexp_split launch: the grid (sized to the residency so that the window tables stay per resident
// group) and the table scratch (table2: a second buffer, for a half that runs beside another)
static int prepare_split(phe_hip_ctx* ctx, const DevSplit& M, const DevSchedule& E, const uint32_t* base, int base_limbs,
                         const uint32_t* post, int post_limbs, uint32_t* out, int out_limbs, size_t batch, int per_cu,
                         bool second_table, SplitArgs& A, int& blocks) {
    if (per_cu < 0) return fail(PHE_HIP_EINVAL, "unsupported split geometry");
    blocks = grid_blocks(ctx, batch, M.G, per_cu);
    const size_t rows = (size_t)blocks * (size_t)(kBlock / M.G);
    uint32_t** tbl = second_table ? &ctx->table2 : &ctx->table;
    int rc = ensure_words(tbl, second_table ? &ctx->table2_words : &ctx->table_words, rows * (size_t)E.tbl_entries * 2 * M.H);
    if (rc) return rc;
    A = SplitArgs{};  // (exit_mod stays null: modexp_split_body takes the quick way out only where a caller sets it)
    A.mod = M.c;
    A.sched = E.ops;
    A.n_ops = E.n_ops;
    A.first_idx = E.first_idx;
    A.tbl_entries = E.tbl_entries;
    A.base = base;
    A.base_limbs = base_limbs;
    A.base_chunks = chunks_for(base_limbs, M.rows);
    A.post = post;
    A.post_limbs = post_limbs;
    A.post_chunks = chunks_for(post_limbs, M.rows);
    A.out = out;
    A.out_limbs = out_limbs;
    A.table = *tbl;
    A.batch = batch;
    A.item_meta = nullptr;
    return PHE_HIP_OK;
}

template <int MODE>
static int launch_split(phe_hip_ctx* ctx, const DevSplit& M, const DevSchedule& E, const uint32_t* base, int base_limbs,
                        const uint32_t* post, int post_limbs, uint32_t* out, int out_limbs, size_t batch,
                        hipStream_t stream, bool second_table = false, bool unit = false) {
    int per_cu = ctx->blocks_per_cu;
    if (per_cu == 0) per_cu = unit ? PHE_SPLIT_BY_GROUP(M.G, occ_split_unit(M.L)) : PHE_SPLIT_BY_GROUP(M.G, occ_split(M.L, MODE));
    SplitArgs A;
    int blocks = 0;
    if (int rc = prepare_split(ctx, M, E, base, base_limbs, post, post_limbs, out, out_limbs, batch, per_cu, second_table, A, blocks))
        return rc;
    if ((unit ? PHE_SPLIT_BY_GROUP(M.G, launch_split_unit(M.L, blocks, stream, A))
              : PHE_SPLIT_BY_GROUP(M.G, launch_split(M.L, MODE, blocks, stream, A))) < 0)
        return fail(PHE_HIP_EINVAL, "unsupported split geometry");
    HIP_TRY(hipGetLastError());
    return PHE_HIP_OK;
}

// both half-exponentiations (moduli Mp, Mq of the same geometry) as one grid: k_modexp_split_halves
static int launch_split_halves(phe_hip_ctx* ctx, const DevSplit& Mp, const DevSchedule& Ep, const DevSplit& Mq, const DevSchedule& Eq,
                               const uint32_t* base, int base_limbs, uint32_t* out_p, uint32_t* out_q, int out_limbs, size_t batch,
                               hipStream_t stream) {
    if (Mp.G != Mq.G || Mp.L != Mq.L) return fail(PHE_HIP_EINVAL, "the two halves need one geometry");
    int per_cu = ctx->blocks_per_cu;
    if (per_cu == 0) per_cu = PHE_SPLIT_BY_GROUP(Mp.G, occ_split(Mp.L, kModeHalfDecrypt));
    SplitArgs Ap, Aq;
    int bp = 0, bq = 0;
    // each half gets half of the residency: together they are one kernel's worth of resident workgroups
    const int half_cu = per_cu > 1 ? per_cu / 2 : per_cu;
    int rc = prepare_split(ctx, Mp, Ep, base, base_limbs, nullptr, 0, out_p, out_limbs, batch, half_cu, false, Ap, bp);
    if (!rc) rc = prepare_split(ctx, Mq, Eq, base, base_limbs, nullptr, 0, out_q, out_limbs, batch, half_cu, true, Aq, bq);
    if (rc) return rc;
    if (PHE_SPLIT_BY_GROUP(Mp.G, launch_split_halves(Mp.L, bp, stream, Ap, Aq)) < 0) return fail(PHE_HIP_EINVAL, "unsupported split geometry");
    HIP_TRY(hipGetLastError());
    return PHE_HIP_OK;
}

// ---- the late single-wave kernels (split_core.h modexp_split_late_body): rungs of 16 lanes and of the whole wave ---------------
// offered where the rung carries the two constant sets (key_setup.h QuickPack: the scaled modulus and, for the way out, the true
// one) and a part holds the kernel for its lane width; PHE_HIP_NO_LATE=1 keeps the round-3 kernels (A/B measurements, tests)
// Two kernels behind the same constants: the whole wave runs the late sweeps (k_modexp_split_late), 16-lane groups the textbook
// order on the scaled modulus with the quick way out (k_modexp_split<..., unit> with A.exit_mod set: "quick").
static int occ_small_rung(const DevSplit& M, int mode) {
    return M.G == 64 ? PHE_SPLIT_BY_GROUP(64, occ_split_late(M.q_L, mode)) : PHE_SPLIT_BY_GROUP(M.G, occ_split_quick(M.q_L, mode));
}
static bool late_offered(const phe_hip_ctx* ctx, const DevSplit& M) {
    return !ctx->no_late && ctx->use_split && (M.G == 16 || M.G == 64) && M.q_L != 0 && M.q_L <= host::kMaxLateL &&
           occ_small_rung(M, kModeEncrypt) > 0;
}
static int prepare_late(phe_hip_ctx* ctx, const DevSplit& M, const DevSchedule& E, const uint32_t* base, int base_limbs,
                        const uint32_t* post, int post_limbs, uint32_t* out, int out_limbs, size_t batch, int per_cu, bool second_table,
                        SplitArgs& A, int& blocks) {
    if (per_cu < 0) return fail(PHE_HIP_EINVAL, "no late kernel for this geometry");
    // the way out modulo the true modulus (A.exit_mod) is compiled into the 16-lane and whole-wave kernels only
    // (split_core.h modexp_split_body): a narrower rung launched with it would return residues of the scaled modulus
    if (M.G < 16) return fail(PHE_HIP_EINVAL, "the quick way out needs groups of 16 lanes or the whole wave");
    blocks = grid_blocks(ctx, batch, M.G, per_cu);
    const size_t rows = (size_t)blocks * (size_t)(kBlock / M.G);
    uint32_t** tbl = second_table ? &ctx->table2 : &ctx->table;
    int rc = ensure_words(tbl, second_table ? &ctx->table2_words : &ctx->table_words, rows * (size_t)E.tbl_entries * 2 * M.q_H);
    if (rc) return rc;
    A = SplitArgs{};
    A.mod = M.q_scaled;
    A.exit_mod = M.q_exit;
    A.sched = E.ops;
    A.n_ops = E.n_ops;
    A.first_idx = E.first_idx;
    A.tbl_entries = E.tbl_entries;
    A.base = base;
    A.base_limbs = base_limbs;
    A.base_chunks = chunks_for(base_limbs, M.q_rows);
    A.post = post;
    A.post_limbs = post_limbs;
    A.post_chunks = chunks_for(post_limbs, M.q_rows);
    A.out = out;
    A.out_limbs = out_limbs;
    A.table = *tbl;
    A.batch = batch;
    A.item_meta = nullptr;
    return PHE_HIP_OK;
}
static int launch_late(phe_hip_ctx* ctx, int mode, const DevSplit& M, const DevSchedule& E, const uint32_t* base, int base_limbs,
                       const uint32_t* post, int post_limbs, uint32_t* out, int out_limbs, size_t batch, hipStream_t stream) {
    int per_cu = ctx->blocks_per_cu;
    if (per_cu == 0) per_cu = occ_small_rung(M, mode);
    SplitArgs A;
    int blocks = 0;
    if (int rc = prepare_late(ctx, M, E, base, base_limbs, post, post_limbs, out, out_limbs, batch, per_cu, false, A, blocks)) return rc;
    if ((M.G == 64 ? PHE_SPLIT_BY_GROUP(64, launch_split_late(M.q_L, mode, blocks, stream, A))
                   : PHE_SPLIT_BY_GROUP(M.G, launch_split_quick(M.q_L, mode, blocks, stream, A))) < 0)
        return fail(PHE_HIP_EINVAL, "no late kernel for this geometry");
    HIP_TRY(hipGetLastError());
    return PHE_HIP_OK;
}
static int launch_late_halves(phe_hip_ctx* ctx, const DevSplit& Mp, const DevSchedule& Ep, const DevSplit& Mq, const DevSchedule& Eq,
                              const uint32_t* base, int base_limbs, uint32_t* out_p, uint32_t* out_q, int out_limbs, size_t batch,
                              hipStream_t stream) {
    if (Mp.G != Mq.G || Mp.q_L != Mq.q_L) return fail(PHE_HIP_EINVAL, "the two halves need one geometry");
    int per_cu = ctx->blocks_per_cu;
    if (per_cu == 0) per_cu = occ_small_rung(Mp, kModeHalfDecrypt);
    SplitArgs Ap, Aq;
    int bp = 0, bq = 0;
    const int half_cu = per_cu > 1 ? per_cu / 2 : per_cu;  // together the halves are one kernel's worth of resident workgroups
    int rc = prepare_late(ctx, Mp, Ep, base, base_limbs, nullptr, 0, out_p, out_limbs, batch, half_cu, false, Ap, bp);
    if (!rc) rc = prepare_late(ctx, Mq, Eq, base, base_limbs, nullptr, 0, out_q, out_limbs, batch, half_cu, true, Aq, bq);
    if (rc) return rc;
    if ((Mp.G == 64 ? PHE_SPLIT_BY_GROUP(64, launch_split_late_halves(Mp.q_L, bp, stream, Ap, Aq))
                    : PHE_SPLIT_BY_GROUP(Mp.G, launch_split_quick_halves(Mp.q_L, bp, stream, Ap, Aq))) < 0)
        return fail(PHE_HIP_EINVAL, "no late kernel for this geometry");
    HIP_TRY(hipGetLastError());
    return PHE_HIP_OK;
}

// One number on a pair of wavefronts (k_modexp_split_ab, whole-wave geometry only): `halves` == 2 runs Mp/Ep on the even
// workgroups and Mq/Eq on the odd ones (the two CRT halves of a decrypt), else Mp/Ep alone.  The window tables are per NUMBER
// here (tbl_entries + 1 pairs each), in ctx->table / ctx->table2.
static bool ab_offered(const phe_hip_ctx* ctx, const DevSplit& M, size_t batch, int halves) {
    // while every wave of every number still finds a SIMD of its own (2 roles x halves x batch waves) — two waves per SIMD for
    // the one-limb-per-lane sweeps, whose dependent chain leaves half the issue slots free: 512 decrypts at 2048 bits 305 k/s
    // against 198 k/s on the single-wave kernels, while two limbs per lane lose 19 % that way
    // (profiles/r03g_wave_pairs_beyond_one_wave_per_simd.txt; PHE_HIP_WAVE_PAIR_DEPTH multiplies the bound, for measurements)
    const size_t per_simd = (size_t)(M.q_L == 1 ? 2 : 1) * (size_t)ctx->wave_pair_depth;
    return !ctx->no_wave_pairs && M.G == 64 && M.q_L != 0 && batch * 2 * (size_t)halves <= (size_t)ctx->n_cus * 4 * per_simd;
}
static int launch_split_ab(phe_hip_ctx* ctx, int mode, const DevSplit& Mp, const DevSchedule& Ep, const DevSplit* Mq, const DevSchedule* Eq,
                           const uint32_t* base, int base_limbs, const uint32_t* post, int post_limbs, uint32_t* out_p, uint32_t* out_q,
                           int out_limbs, size_t batch, hipStream_t stream, const uint32_t* item_meta = nullptr) {
    const int halves = Mq ? 2 : 1;
    if (Mp.q_L == 0 || (Mq && Mq->q_L != Mp.q_L)) return fail(PHE_HIP_EINVAL, "the two halves need one wave-pair geometry");
    SplitArgs Ap, Aq;
    const auto fill = [&](SplitArgs& A, const DevSplit& M, const DevSchedule& E, uint32_t* out, bool second) -> int {
        uint32_t** tbl = second ? &ctx->table2 : &ctx->table;
        int rc = ensure_words(tbl, second ? &ctx->table2_words : &ctx->table_words, batch * (size_t)(E.tbl_entries + 1) * 2 * M.q_H);
        if (rc) return rc;
        A.mod = M.q_scaled;
        A.exit_mod = M.q_exit;
        A.sched = E.ops;
        A.n_ops = E.n_ops;
        A.first_idx = E.first_idx;
        A.tbl_entries = E.tbl_entries;
        A.base = base;
        A.base_limbs = base_limbs;
        A.base_chunks = chunks_for(base_limbs, M.q_rows);
        A.post = post;
        A.post_limbs = post_limbs;
        A.post_chunks = chunks_for(post_limbs, M.q_rows);
        A.out = out;
        A.out_limbs = out_limbs;
        A.table = *tbl;
        A.batch = batch;
        A.item_meta = item_meta;
        return PHE_HIP_OK;
    };
    int rc = fill(Ap, Mp, Ep, out_p, false);
    if (!rc && Mq) rc = fill(Aq, *Mq, *Eq, out_q, true);
    if (rc) return rc;
    if (!Mq) Aq = Ap;
    if (PHE_SPLIT_BY_GROUP(64, launch_split_ab(Mp.q_L, mode, (int)batch, halves, stream, Ap, Aq)) < 0)
        return fail(PHE_HIP_EINVAL, "no wave-pair kernel for this geometry");
    HIP_TRY(hipGetLastError());
    return PHE_HIP_OK;
}

static int launch_var_split(phe_hip_ctx* ctx, const DevSplit& M, const uint32_t* base, int base_limbs,
                            const uint32_t* e, int exp_limbs, int max_bits, uint32_t* out, int out_limbs, size_t batch,
                            hipStream_t stream, bool pair_io = false) {
    SplitVarArgs A;
    A.pair_io = pair_io ? 1 : 0;
    A.mod = M.c;
    A.base = base;
    A.base_limbs = base_limbs;
    A.base_chunks = chunks_for(base_limbs, M.rows);
    A.exps = e;
    A.exp_limbs = exp_limbs;
    A.window = host::pick_window(max_bits);
    A.n_windows = std::max(1, (max_bits + A.window - 1) / A.window);
    A.out = out;
    A.out_limbs = out_limbs;
    A.batch = batch;
    int per_cu = ctx->blocks_per_cu;
    if (per_cu == 0) per_cu = PHE_SPLIT_BY_GROUP(M.G, occ_var_split(M.L));
    if (per_cu < 0) return fail(PHE_HIP_EINVAL, "unsupported split geometry");
    const int blocks = grid_blocks(ctx, batch, M.G, per_cu);
    const size_t rows = (size_t)blocks * (size_t)(kBlock / M.G);
    int rc = ensure_words(&ctx->table, &ctx->table_words, rows * ((size_t)1 << A.window) * 2 * M.H);
    if (rc) return rc;
    A.table = ctx->table;
    if (PHE_SPLIT_BY_GROUP(M.G, launch_var_split(M.L, blocks, stream, A)) < 0)
        return fail(PHE_HIP_EINVAL, "unsupported split geometry");
    HIP_TRY(hipGetLastError());
    return PHE_HIP_OK;
}


static int launch_multi_split(phe_hip_ctx* ctx, const DevSplit& M, const uint32_t* base, const uint32_t* base_inv,
                              int base_limbs, const uint32_t* e, const uint8_t* neg, int exp_limbs, int max_bits, int chunk,
                              int row_block, uint32_t* out, int out_limbs, size_t batch, size_t rows, hipStream_t stream,
                              bool pair_in = false) {
    SplitMultiArgs A;
    A.pair_in = pair_in ? 1 : 0;
    A.mod = M.c;
    A.base = base;
    A.base_inv = base_inv;
    A.base_limbs = base_limbs;
    A.base_chunks = chunks_for(base_limbs, M.rows);
    A.exps = e;
    A.neg = neg;
    A.exp_limbs = exp_limbs;
    A.window = host::pick_multi_window(max_bits);
    A.n_windows = std::max(1, (max_bits + A.window - 1) / A.window);
    A.chunk = chunk;
    A.row_block = row_block;
    A.out = out;
    A.out_limbs = out_limbs;
    A.batch = batch;
    A.rows = rows;
    A.n_chunks = (batch + (size_t)chunk - 1) / (size_t)chunk;
    A.n_row_blocks = (rows + (size_t)row_block - 1) / (size_t)row_block;
    int per_cu = ctx->blocks_per_cu;
    if (per_cu == 0) per_cu = PHE_SPLIT_BY_GROUP(M.G, occ_multi_split(M.L));
    if (per_cu < 0) return fail(PHE_HIP_EINVAL, "unsupported split geometry");
    const int blocks = grid_blocks(ctx, (size_t)(A.n_chunks * A.n_row_blocks), M.G, per_cu);
    const size_t groups = (size_t)blocks * (size_t)(kBlock / M.G);
    int rc = ensure_words(&ctx->table, &ctx->table_words,
                          groups * (size_t)chunk * (((size_t)1 << A.window) - 1) * (base_inv ? 2 : 1) * 2 * M.H);
    if (rc) return rc;
    A.table = ctx->table;
    if (PHE_SPLIT_BY_GROUP(M.G, launch_multi_split(M.L, blocks, stream, A)) < 0)
        return fail(PHE_HIP_EINVAL, "unsupported split geometry");
    HIP_TRY(hipGetLastError());
    return PHE_HIP_OK;
}

static int launch_var(phe_hip_ctx* ctx, const DevModulus& M, const uint32_t* base, int base_limbs,
                      const uint32_t* e, int exp_limbs, int max_bits, uint32_t* out, int out_limbs, size_t batch,
                      hipStream_t stream) {
    VarArgs A;
    A.mod = M.c;
    A.base = base;
    A.base_limbs = base_limbs;
    A.exps = e;
    A.exp_limbs = exp_limbs;
    A.window = host::pick_window(max_bits);
    A.n_windows = std::max(1, (max_bits + A.window - 1) / A.window);
    A.out = out;
    A.out_limbs = out_limbs;
    A.batch = batch;
    int G, L;
    light_geometry(M, G, L);
    int per_cu = ctx->blocks_per_cu;
    if (per_cu == 0) per_cu = PHE_BY_GROUP(G, occ_var(L));
    if (per_cu < 0) return fail(PHE_HIP_EINVAL, "unsupported limb-group geometry");
    const int blocks = grid_blocks(ctx, batch, G, per_cu);
    const size_t rows = (size_t)blocks * (size_t)(kBlock / G);
    int rc = ensure_words(&ctx->table, &ctx->table_words, rows * ((size_t)1 << A.window) * M.S);
    if (rc) return rc;
    A.table = ctx->table;
    if (PHE_BY_GROUP(G, launch_var(L, blocks, stream, A)) < 0) return fail(PHE_HIP_EINVAL, "unsupported limb-group geometry");
    HIP_TRY(hipGetLastError());
    return PHE_HIP_OK;
}

// batches from this many rows on take the table kernel: one 512-thread workgroup per CU (32 limb groups) copies the 80 KB table
// into LDS before its first product — measured against the Montgomery kernels on one box (profiles/r04g_*): 2^11 rows 71 M against
// 87 M/s, 2^12 140 against 171, 2^13 276 against 212, 2^20 368 against 313
static const size_t kTableMulMinRows = 8192;
// ... and from this many rows on the tile kernel (mul_tile.h: one element per lane, a 1024-thread workgroup per CU takes 64 products
// at a time): measured on one box (profiles/r04k_*, r04q_*), M products/s at 2^13 / 2^14 / 2^15 / 2^20 rows: two Montgomery
// products 211 / 264 / 287 / 303, table in LDS 276 / 312 / 328 / 363, tiles 184 / 338 / 377 / 429
static const size_t kTileMulMinRows = 16384;
static int launch_mul(phe_hip_ctx* ctx, const DevModulus& M, const uint32_t* a, size_t a_stride, const uint32_t* b,
                      size_t b_stride, uint32_t* out, size_t out_stride, int limbs, size_t batch,
                      hipStream_t stream, int b_plain_limbs = 0, int one_product = 0, int a_limbs = 0, bool plain_mulmod = false) {
    static const char* const min_rows_env = getenv("PHE_HIP_TABLE_MUL_MIN_ROWS");  // (measurements of the crossover; read once)
    static const size_t min_rows = min_rows_env ? (size_t)std::max(1, atoi(min_rows_env)) : kTableMulMinRows;
    static const size_t tile_min_rows = min_rows_env ? min_rows : kTileMulMinRows;
    if (plain_mulmod && ctx->tmul_blob && !b_plain_limbs && !one_product && !a_limbs && batch >= min_rows && limbs == ctx->pub.s2 &&
        limbs % 4 == 0 && ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15u) == 0) && a_stride % 4 == 0 && b_stride % 4 == 0 &&
        out_stride % 4 == 0 && b_stride != 0) {
        // _raw_add on plain residues (phe/paillier.py:705-719): one plain product + one fold against the key's table
        // (mul_table.h) instead of two Montgomery products
        const host::TableMulPack& T = ctx->tmul;
        TableMulArgs B;
        B.n = ctx->tmul_blob;
        B.ncomp = ctx->tmul_blob + T.S;
        B.ncomp1 = ctx->tmul_blob + 2 * T.S;
        B.table = ctx->tmul_blob + 3 * T.S;
        B.inv = T.inv;
        B.split = T.split;
        B.digits = T.digits;
        B.base = T.base;
        B.a = a;
        B.b = b;
        B.out = out;
        B.a_stride = a_stride;
        B.b_stride = b_stride;
        B.out_stride = out_stride;
        B.limbs = limbs;
        B.batch = batch;
        B.digits_padded = T.digits_padded;
        B.tile_waves = T.tile_waves;
        int rc = -1;
        if (ctx->tmul_cols && batch >= tile_min_rows) {
            // by tiles of 64 products per workgroup, the fold on one element per lane with the table words on the scalar path
            // (mul_tile.h); PHE_HIP_NO_TILE_MUL=1 keeps the kernel with the table in LDS
            TableMulArgs C = B;
            C.table = ctx->tmul_cols;
            const size_t tiles = (batch + 63) / 64;
            const size_t slots = (size_t)ctx->n_cus * (size_t)phe::t16::tile_blocks_per_cu(T.tile_waves);
            rc = phe::t16::launch_mul_tile(T.L, (int)std::max<size_t>(1, std::min(tiles, slots)), stream, C);
            if (rc == 0) ctx->last_path |= kPathTileMul;
        }
        if (rc != 0 && T.in_lds()) {
            (void)hipGetLastError();
            const size_t per_block = 32;  // limb groups of a 512-thread workgroup
            const int blocks = (int)std::max<size_t>(1, std::min((batch + per_block - 1) / per_block, (size_t)ctx->n_cus));
            rc = phe::t16::launch_mul_table(T.L, blocks, T.lds_words * 4, stream, B);
        }
        if (rc == 0) {
            HIP_TRY(hipGetLastError());
            ctx->last_path |= kPathTableMul;
            return PHE_HIP_OK;
        }
        (void)hipGetLastError();  // (-1 / -2: no kernel for the width, or the LDS size refused: the Montgomery kernels serve)
    }
    MulArgs A;
    A.b_plain_limbs = b_plain_limbs;
    A.one_product = one_product;
    A.a_limbs = a_limbs;
    // 16-byte chunks (mul_io.h) when every row starts on a 16-byte boundary; otherwise word by word
    A.vec_ok = ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15u) == 0 && a_stride % 4 == 0 && b_stride % 4 == 0 &&
                out_stride % 4 == 0 && limbs % 4 == 0 && b_plain_limbs % 4 == 0 && a_limbs % 4 == 0) ? 1 : 0;
    A.mod = M.c;
    A.a = a;
    A.b = b;
    A.out = out;
    A.a_stride = a_stride;
    A.b_stride = b_stride;
    A.out_stride = out_stride;
    A.limbs = limbs;
    A.batch = batch;
    if (M.G == 0) {
        // n^2 is wider than the widest full-width geometry (keys above ~4170 bits): the product runs on the pair form
        const DevSplit& SP = ctx->d_nsplit;
        if (!SP.G || one_product) return fail(PHE_HIP_EINVAL, "no product kernel for this key width");
        SplitMulArgs B;
        B.mod = SP.c;
        B.a = a;
        B.b = b;
        B.out = out;
        B.a_stride = a_stride;
        B.b_stride = b_stride;
        B.out_stride = out_stride;
        B.limbs = limbs;
        B.chunks = chunks_for(limbs, SP.rows);
        B.b_plain_limbs = b_plain_limbs;
        B.batch = batch;
        const int blocks = grid_blocks(ctx, batch, SP.G, 2);
        if (PHE_SPLIT_BY_GROUP(SP.G, launch_mul_split(SP.L, blocks, stream, B)) < 0)
            return fail(PHE_HIP_EINVAL, "unsupported split geometry for the product");
        HIP_TRY(hipGetLastError());
        return PHE_HIP_OK;
    }
    // no table scratch here: let every CU hold as many groups as the batch offers (cap 8 blocks/CU)
    int G, L;
    light_geometry(M, G, L);
    const int blocks = grid_blocks(ctx, batch, G, 8);
    if (PHE_BY_GROUP(G, launch_mul(L, blocks, stream, A)) < 0) return fail(PHE_HIP_EINVAL, "unsupported limb-group geometry");
    HIP_TRY(hipGetLastError());
    return PHE_HIP_OK;
}

// ---- the geometry ladder --------------------------------------------------------------------------------------------
// A launch of `batch` numbers on groups of G lanes occupies batch*G lanes; the chip offers n_cus * 4 SIMDs * 64 lanes per wave
// slot.  A product is a chain of `rows` dependent digit steps of ~(19 L + 52) issue cycles each (4L multiply-adds at ~4.8
// cycles + ~13 other instructions), so a rung's time for one residency of waves is rows * (19 L + 52), and a batch
// that needs w waves per SIMD takes max(1, w) of those: the rung with the least estimated time wins.  One wave per SIMD runs
// these kernels at ~95 % of their two-wave rate, a SIMD without a wave runs nothing — narrow groups (most limbs per lane,
// fewest instructions per multiply-add) win as soon as they fill the chip, wide ones (short chains) below that; measured
// crossovers: profiles/r03*_batch_sweep*.
struct RungShape {
    int G = 0, L = 0, rows = 0;  // G == 0: this rung has no kernel of the family asked for
    double wide = 1.0;           // what a rung too wide for the fused sweeps costs on top of the estimate (split_shape)
    int family = 0;              // RungFamily: which measured table serves (measured_cost)
};
// Rungs with more than kMaxFusedL limbs per lane (3072-bit keys: 4 x 27 and, for the CRT halves, 2 x 27) run every pair
// product as two single sweeps with the quotient digits handed over through LDS, and the estimate above is too kind to them
// — by how much depends on the kernel family.  Measured at 3072 bits against the next rung pinned for every batch size
// (profiles/r03r_pinned_rungs_3072.txt): the fixed-exponent ladder (encrypt, obfuscate) is still 6 % FASTER on 4 x 27 than
// on 8 x 14 once two waves per SIMD are there (factor 1.10 on the estimate), the CRT halves are 7 % slower on 2 x 27 than on
// 4 x 14 and the per-element exponents 7 % slower on 4 x 27 than on 8 x 14 (factor 1.25); with ONE wave per SIMD those
// kernels reach 86 % of their two-wave rate where the fused ones reach 90-95 % (factor 1.16 while w <= 1).
// (kFamHalvesLongExp: the CRT halves with the exponent n — the key owner's encryption.  Same kernels and the same wide-rung factor as
//  kFamHalves, but no measured table serves it: family 2's launch times were taken with the decrypt's exponents, half as long)
enum RungFamily : int { kFamOther = 0, kFamFixedExp = 1, kFamHalves = 2, kFamVarExp = 3, kFamHalvesLongExp = 4 };
static double wide_rung_factor(int family) {
    switch (family) {
        case kFamFixedExp: return 1.10;
        case kFamHalves:
        case kFamHalvesLongExp:
        case kFamVarExp: return 1.25;
        default: return 1.0;  // not measured: the plain estimate, as before
    }
}
static double rung_cost(const phe_hip_ctx* ctx, size_t batch, const RungShape& r, int concurrent) {
    const double lanes = (double)ctx->n_cus * 256.0;
    const double w = (double)batch * r.G * concurrent / lanes;
    const bool wide = r.L > kMaxFusedL && r.wide > 1.0;
    const double residencies = (wide && w <= 1.0) ? 1.16 : std::max(1.0, w);
    return (double)r.rows * (19.0 * r.L + 52.0) * (wide ? r.wide : 1.0) * residencies;
}
// The MEASURED cost of a rung: the launch time (ns) of `batch` numbers of family r.family on groups of r.G lanes, from the table the
// context was given (phe_hip_ctx_load_ladder) — piecewise linear in the batch size between the measured sizes; below the smallest
// one the launch is latency-bound (its time stays), above the largest one it scales with the rows.  < 0: no table for this rung.
static double measured_cost(const phe_hip_ctx* ctx, size_t batch, const RungShape& r) {
    const auto it = ctx->ladder.find(r.family * 256 + r.G);
    if (it == ctx->ladder.end() || it->second.empty()) return -1.0;
    const std::vector<phe_hip_ctx::LadderPoint>& p = it->second;
    const double b = (double)batch;
    if (b <= p.front().rows) return p.front().ns;
    if (b >= p.back().rows) return p.back().ns * b / p.back().rows;
    size_t hi = 1;
    while (p[hi].rows < b) ++hi;
    const double t = (b - p[hi - 1].rows) / (p[hi].rows - p[hi - 1].rows);
    return p[hi - 1].ns + t * (p[hi].ns - p[hi - 1].ns);
}
// rung index (0 = the members of the context, k >= 1 = the k-th extra rung): least time for this batch — measured when the
// context holds a table for EVERY candidate rung of the family (the two kinds of cost are not comparable), estimated otherwise
template <class Shape>
static int pick_rung(const phe_hip_ctx* ctx, size_t batch, int n_rungs, Shape shape, int concurrent = 1) {
    if (ctx->force_group) {
        for (int k = 0; k < n_rungs; ++k)
            if (shape(k).G >= ctx->force_group) return k;
        for (int k = n_rungs - 1; k >= 0; --k)
            if (shape(k).G) return k;
        return 0;
    }
    // (the tables are keyed by family and group width alone: launch times taken with the default kernels must not steer the
    //  variants the measurement switches select — PHE_HIP_NO_LATE, PHE_HIP_FORCE_UNIT, a blocks-per-CU override)
    bool measured = !ctx->ladder.empty() && !ctx->no_late && !ctx->force_unit && ctx->blocks_per_cu == 0;
    for (int k = 0; measured && k < n_rungs; ++k) {
        const RungShape r = shape(k);
        if (r.G && measured_cost(ctx, batch, r) < 0) measured = false;
    }
    int best = -1;
    double best_cost = 0;
    for (int k = 0; k < n_rungs; ++k) {
        const RungShape r = shape(k);
        if (!r.G) continue;
        const double c = measured ? measured_cost(ctx, batch, r) : rung_cost(ctx, batch, r, concurrent);
        if (best < 0 || c < best_cost * 0.97) {  // a wider rung must be clearly better: ties go to the narrower one
            best = k;
            best_cost = c;
        }
    }
    return best < 0 ? 0 : best;
}
static int light_group(const DevModulus& M) {
    int G, L;
    light_geometry(M, G, L);
    return G;
}
static const DevModulus& nsq_rung(const phe_hip_ctx* ctx, int k) { return k == 0 ? ctx->d_nsq : ctx->pub_rungs[(size_t)k - 1].nsq; }
static const DevSplit& nsplit_rung(const phe_hip_ctx* ctx, int k) { return k == 0 ? ctx->d_nsplit : ctx->pub_rungs[(size_t)k - 1].nsplit; }
// the full-width kernels modulo n^2 (products; the second engine)
static const DevModulus& pick_nsq(const phe_hip_ctx* ctx, size_t batch) {
    const int k = pick_rung(ctx, batch, 1 + (int)ctx->pub_rungs.size(), [&](int r) {
        RungShape sh;
        const DevModulus& M = nsq_rung(ctx, r);
        if (M.G) {
            light_geometry(M, sh.G, sh.L);
            sh.rows = M.S;
            // lanes of more than 18 limbs take the plain product body (no LDS-DMA staging: mul_io.h RowIO::kUse) — measured at 3072
            // bits: 8 x 27 limbs 117 M products/s, 16 x 14 (staged) 138 M/s at every batch size (profiles/r04e_*)
            sh.wide = 1.5;
        }
        return sh;
    });
    return nsq_rung(ctx, k);
}
// late: the rung runs this family on the late sweeps (late_offered): rows = the limbs the scaled modulus really needs + 1, not
// the G*L the lanes could hold — e.g. 40 instead of 48 for the CRT halves of a 2048-bit key on 16 lanes x 3 limbs
static RungShape split_shape(const DevSplit& sp, int family = kFamOther, bool late = false) {
    RungShape sh;
    sh.G = sp.G;
    sh.L = late ? sp.q_L : sp.L;
    sh.rows = late ? sp.q_rows + (sp.G == 64 ? 1 : 0) : sp.rows;
    sh.wide = wide_rung_factor(family);
    sh.family = family;
    return sh;
}
// the pair-form kernels modulo n
static bool late_offered(const phe_hip_ctx* ctx, const DevSplit& M);
static int pick_nsplit_rung(const phe_hip_ctx* ctx, size_t batch, int family = kFamOther) {
    return pick_rung(ctx, batch, 1 + (int)ctx->pub_rungs.size(), [&](int r) {
        const DevSplit& sp = nsplit_rung(ctx, r);
        return split_shape(sp, family, family == kFamFixedExp && late_offered(ctx, sp));
    });
}
static const DevSplit& pick_nsplit(const phe_hip_ctx* ctx, size_t batch, int family = kFamOther) {
    return nsplit_rung(ctx, pick_nsplit_rung(ctx, batch, family));
}
static int geom_code(int G, int L) { return G * 100 + L; }

static int check_ctx(const phe_hip_ctx* ctx) {
    if (!ctx) return fail(PHE_HIP_EINVAL, "null context");
    return PHE_HIP_OK;
}
static int bind_device(const phe_hip_ctx* ctx) {
    HIP_TRY(hipSetDevice(ctx->device));
    return PHE_HIP_OK;
}

// One stream order per context.  The window tables, the decrypt intermediates and the staging blocks belong to the context,
// not to a call: two calls issued on different streams would otherwise run concurrently on the same scratch.  Every entry
// point that launches on the context's scratch therefore makes its stream wait for the previous call's work (an event; no
// host synchronisation) and leaves its own event behind.  Calls from several HOST threads still need the caller's lock.
struct CtxOrder {
    phe_hip_ctx* ctx;
    hipStream_t st;
    int rc = PHE_HIP_OK;
    // Every call leaves an event behind on ITS OWN stream when it returns (everything it queued is covered); a call that
    // arrives on another stream waits for that event.  No handle of a foreign stream is kept beyond the call that was given
    // it: a caller may destroy its stream as soon as the call has returned (recording on a destroyed stream is undefined
    // behaviour — round 3 recorded lazily on the PREVIOUS call's stream).
    CtxOrder(phe_hip_ctx* c, void* stream) : ctx(c), st((hipStream_t)stream) {
        if (!ctx->busy_valid || ctx->busy_stream == st) return;
        hipError_t e = hipStreamWaitEvent(st, ctx->ev_busy, 0);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            e = hipDeviceSynchronize();  // whatever ran before is ordered by a full drain
            if (e != hipSuccess) rc = fail(PHE_HIP_EHIP, std::string("stream order of the context: ") + hipGetErrorString(e));
        }
    }
    ~CtxOrder() {
        hipError_t e = hipSuccess;
        if (!ctx->ev_busy) e = hipEventCreateWithFlags(&ctx->ev_busy, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(ctx->ev_busy, st);
        if (e != hipSuccess) {  // no event: the next call on another stream must not run beside this one
            (void)hipGetLastError();
            (void)hipStreamSynchronize(st);
        }
        ctx->busy_stream = st;  // (compared only, never used as a handle again)
        ctx->busy_valid = e == hipSuccess;
    }
};
#define PHE_CTX_ORDER(ctx, stream)   \
    CtxOrder order_(ctx, stream);    \
    if (order_.rc) return order_.rc

extern "C" {

const char* phe_hip_last_error(void) { return g_err.c_str(); }

int phe_hip_abi_version(void) { return PHE_HIP_ABI_VERSION; }

int phe_hip_device_count(int* count) {
    if (!count) return fail(PHE_HIP_EINVAL, "null count");
    HIP_TRY(hipGetDeviceCount(count));
    return PHE_HIP_OK;
}

static int ctx_common(phe_hip_ctx* ctx, const uint32_t* n, int n_limbs, int device) {
    if (!n || n_limbs < 1) return fail(PHE_HIP_EINVAL, "n / n_limbs invalid");
    ctx->device = device;
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    ctx->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    ctx->arch = prop.gcnArchName;
    if (const char* e = getenv("PHE_HIP_BLOCKS_PER_CU")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 8) ctx->blocks_per_cu = v;
    }
    if (const char* e = getenv("PHE_HIP_ENGINE")) ctx->use_split = (strcmp(e, "full") != 0);
    if (const char* e = getenv("PHE_HIP_GROUP")) {
        const int v = atoi(e);
        if (v == 2 || v == 4 || v == 8 || v == 16) ctx->prefer_group = v;
    }
    try {
        ctx->pub = host::build_public(n, n_limbs, ctx->prefer_group);
    } catch (const std::exception& ex) {
        return fail(PHE_HIP_EINVAL, ex.what());
    }
    if (ctx->pub.nsq.G == 0) ctx->use_split = true;  // wide keys have the pair form only
    int rc = upload_modulus(ctx->pub.nsq, ctx->d_nsq);
    if (!rc) rc = upload_split(ctx->pub.nsplit, ctx->d_nsplit, &ctx->pub.nquick);
    if (!rc && !getenv("PHE_HIP_NO_UNIT")) rc = upload_split(ctx->pub.nunit, ctx->d_nunit);
    if (!rc) rc = upload_schedule(ctx->pub.exp_n, ctx->d_exp_n);
    if (!rc && !getenv("PHE_HIP_NO_TABLE_MUL")) {
        try {
            // the 8-wave tile shape first (n^2 of ~810 ... 1025-bit keys fills its 72 columns: half the multiply-adds of the two
            // Montgomery products these widths ran until round 5; PHE_HIP_NO_TILE8=1 keeps those), then the 16-wave / 16-lane shapes
            if (!getenv("PHE_HIP_NO_TILE8") && !getenv("PHE_HIP_NO_TILE_MUL")) ctx->tmul = host::build_table_mul(ctx->pub.nsq32, ctx->pub.s2, false, 8);
            if (!ctx->tmul.ok()) ctx->tmul = host::build_table_mul(ctx->pub.nsq32, ctx->pub.s2, getenv("PHE_HIP_TABLE_MUL_ANY_WIDTH") != nullptr);
        } catch (const std::exception&) {
            ctx->tmul = host::TableMulPack();
        }
        if (ctx->tmul.ok()) {
            const host::TableMulPack& T = ctx->tmul;
            std::vector<uint32_t> h(T.n);
            h.insert(h.end(), T.ncomp.begin(), T.ncomp.end());
            h.insert(h.end(), T.ncomp1.begin(), T.ncomp1.end());
            if (T.in_lds()) h.insert(h.end(), T.table.begin(), T.table.end());  // (3072-bit keys: the tile kernel only, table in L2)
            // an OPTIONAL fast path: a failed allocation or copy leaves the context on the Montgomery kernels (like a key width
            // build_table_mul does not offer) instead of failing its creation
            const auto put = [](uint32_t*& dptr, const std::vector<uint32_t>& words) {
                if (hipMalloc((void**)&dptr, words.size() * 4) != hipSuccess) {
                    dptr = nullptr;
                    (void)hipGetLastError();
                    return false;
                }
                if (hipMemcpy(dptr, words.data(), words.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
                    (void)hipGetLastError();
                    (void)hipFree(dptr);
                    dptr = nullptr;
                    return false;
                }
                return true;
            };
            if (!put(ctx->tmul_blob, h)) ctx->tmul = host::TableMulPack();
            else if (T.tiles() && !getenv("PHE_HIP_NO_TILE_MUL")) (void)put(ctx->tmul_cols, T.table_cols);  // (without it: the table-in-LDS kernel)
        }
    }
    ctx->force_unit = getenv("PHE_HIP_FORCE_UNIT") != nullptr;
    ctx->no_wave_pairs = getenv("PHE_HIP_NO_WAVE_PAIRS") != nullptr;
    ctx->no_late = getenv("PHE_HIP_NO_LATE") != nullptr;
    if (const char* d = getenv("PHE_HIP_WAVE_PAIR_DEPTH")) ctx->wave_pair_depth = std::max(1, atoi(d));
    if (!rc && !getenv("PHE_HIP_GROUP")) {
        // the wider rungs of the ladder: 4-, 8- and 16-lane groups and the whole wave, where they differ from what is already
        // there (4 lanes: keys whose narrowest geometry is 2 lanes wide — n of 1024 bits is 2 x 18 limbs)
        for (int prefer : {4, 8, 16, 64}) {
            try {
                phe_hip_ctx::PubRung R;
                R.plan = host::build_public(n, n_limbs, prefer);
                const auto same = [&](const host::PublicPlan& o) {
                    return o.nsq.G == R.plan.nsq.G && o.nsq.L == R.plan.nsq.L && o.nsplit.G == R.plan.nsplit.G && o.nsplit.L == R.plan.nsplit.L;
                };
                bool have = same(ctx->pub);
                for (const auto& o : ctx->pub_rungs) have = have || same(o.plan);
                if (have) continue;
                // a wider rung is worth having only if it shortens the product: rows x multiply-adds per row (H * L) must
                // drop by a fifth at least (padding to the compiled limb counts can eat the whole gain)
                const host::PublicPlan& prev = ctx->pub_rungs.empty() ? ctx->pub : ctx->pub_rungs.back().plan;
                const auto chain = [&](const host::PublicPlan& o) {
                    return (ctx->use_split && o.nsplit.G) ? o.nsplit.rows * o.nsplit.L : o.nsq.S * o.nsq.L;
                };
                if (chain(R.plan) * 5 > chain(prev) * 4) continue;
                rc = upload_modulus(R.plan.nsq, R.nsq);
                if (!rc) rc = upload_split(R.plan.nsplit, R.nsplit, &R.plan.nquick);
                if (!rc && R.plan.nunit.G && R.plan.nsplit.G != 64 && !getenv("PHE_HIP_NO_UNIT")) rc = upload_split(R.plan.nunit, R.nunit);
                if (rc) break;
                ctx->pub_rungs.push_back(R);
            } catch (const std::exception&) {
                // no compiled kernel of that width covers this key: the ladder simply has no such rung
            }
        }
    }
    return rc;
}

int phe_hip_ctx_create_public(const uint32_t* n, int n_limbs, int device, phe_hip_ctx** out) {
    if (!out) return fail(PHE_HIP_EINVAL, "null out");
    phe_hip_ctx* ctx = new phe_hip_ctx();
    int rc = ctx_common(ctx, n, n_limbs, device);
    if (rc) {
        phe_hip_ctx_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return PHE_HIP_OK;
}

// Constants of the key owner's encryption (split_core.h:crt_lift_body; key_setup.h:build_owner_lift).  Not offered
// (owner_blob stays null, the public path serves) when the half-exponentiations have no pair form or no split part
// holds the full-width geometry of q^2.
static int setup_owner_encrypt(phe_hip_ctx* ctx) {
    const host::ModulusPack& Q = ctx->priv.qsq;
    if (!ctx->use_split || !ctx->priv.psplit.G || !ctx->priv.qsplit.G || !ctx->pub.nsq.G || Q.G == 0) return PHE_HIP_OK;
    DevModulus shape;
    shape.G = Q.G;
    shape.L = Q.L;
    int G, L;
    light_geometry(shape, G, L);
    if (!host::split_part_holds(G, L)) return PHE_HIP_OK;
    try {
        host::OwnerLift W;
        if (!host::build_owner_lift(ctx->priv.tail.p, ctx->priv.tail.q, Q.S, W)) return PHE_HIP_OK;
        std::vector<uint32_t> blob(W.kr);
        blob.insert(blob.end(), W.nkr.begin(), W.nkr.end());
        blob.insert(blob.end(), W.psq.begin(), W.psq.end());
        HIP_TRY(hipMalloc((void**)&ctx->owner_blob, blob.size() * 4));
        HIP_TRY(hipMemcpy(ctx->owner_blob, blob.data(), blob.size() * 4, hipMemcpyHostToDevice));
        ctx->owner_G = G;
        ctx->owner_L = L;
    } catch (const std::exception& ex) {
        return fail(PHE_HIP_EINVAL, ex.what());
    }
    return PHE_HIP_OK;
}

int phe_hip_ctx_create_private(const uint32_t* n, int n_limbs, const uint32_t* p, const uint32_t* q,
                               const uint32_t* hp, const uint32_t* hq, const uint32_t* p_inverse, int pq_limbs,
                               int device, phe_hip_ctx** out) {
    if (!out) return fail(PHE_HIP_EINVAL, "null out");
    if (!p || !q || !hp || !hq || !p_inverse || pq_limbs < 1) return fail(PHE_HIP_EINVAL, "private key material missing");
    phe_hip_ctx* ctx = new phe_hip_ctx();
    int rc = ctx_common(ctx, n, n_limbs, device);
    if (!rc) {
        try {
            ctx->priv = host::build_private(p, q, hp, hq, p_inverse, pq_limbs, n_limbs, ctx->prefer_group);
            // p*q == n, like phe/paillier.py:218-219
            Big prod = host::big_mul(ctx->priv.tail.p, ctx->priv.tail.q);
            Big nn = host::big_from(n, n_limbs, (int)prod.size() > n_limbs ? (int)prod.size() : n_limbs);
            prod.resize(nn.size(), 0u);
            if (host::big_cmp(prod, nn) != 0) throw std::invalid_argument("given public key does not match the given p and q.");
        } catch (const std::exception& ex) {
            rc = fail(PHE_HIP_EINVAL, ex.what());
        }
    }
    if (!rc) rc = upload_modulus(ctx->priv.psq, ctx->d_psq);
    if (!rc) rc = upload_modulus(ctx->priv.qsq, ctx->d_qsq);
    if (!rc) rc = upload_split(ctx->priv.psplit, ctx->d_psplit, &ctx->priv.pquick);
    if (!rc) rc = upload_split(ctx->priv.qsplit, ctx->d_qsplit, &ctx->priv.qquick);
    if (!rc && !getenv("PHE_HIP_GROUP")) {
        for (int prefer : {2, 4, 8, 16, 64}) {  // (2: only where rung 0 is the one-lane geometry of 1024-bit keys' p, q)
            try {
                phe_hip_ctx::PrivRung R;
                R.plan = host::build_private(p, q, hp, hq, p_inverse, pq_limbs, n_limbs, prefer);
                const auto same = [&](const host::PrivatePlan& o) {
                    return o.psq.G == R.plan.psq.G && o.psq.L == R.plan.psq.L && o.psplit.G == R.plan.psplit.G &&
                           o.psplit.L == R.plan.psplit.L && o.qsplit.G == R.plan.qsplit.G && o.qsplit.L == R.plan.qsplit.L;
                };
                bool have = same(ctx->priv);
                for (const auto& o : ctx->priv_rungs) have = have || same(o.plan);
                if (have) continue;
                const host::PrivatePlan& prev = ctx->priv_rungs.empty() ? ctx->priv : ctx->priv_rungs.back().plan;
                const auto chain = [&](const host::PrivatePlan& o) {
                    return (ctx->use_split && o.qsplit.G) ? o.qsplit.rows * o.qsplit.L : o.qsq.S * o.qsq.L;
                };
                if (chain(R.plan) * 5 > chain(prev) * 4) continue;  // see the public rungs
                rc = upload_modulus(R.plan.psq, R.psq);
                if (!rc) rc = upload_modulus(R.plan.qsq, R.qsq);
                if (!rc) rc = upload_split(R.plan.psplit, R.psplit, &R.plan.pquick);
                if (!rc) rc = upload_split(R.plan.qsplit, R.qsplit, &R.plan.qquick);
                if (rc) break;
                ctx->priv_rungs.push_back(R);
            } catch (const std::exception&) {
            }
        }
    }
    if (!rc) rc = upload_schedule(ctx->priv.exp_p, ctx->d_exp_p);
    if (!rc) rc = upload_schedule(ctx->priv.exp_q, ctx->d_exp_q);
    if (!rc) rc = upload_tail(ctx->priv.tail, ctx->d_tail);
    if (!rc && !getenv("PHE_HIP_NO_WAVE_TAIL")) rc = upload_tail_wave(ctx);
    // Which batches take it.  The per-thread tail is bound by its LDS traffic (~9 h^2 word accesses per ciphertext, h = words
    // of p: 8.9 ns per ciphertext at 2048-bit keys, and a 0.38 ms serial chain however small the batch); the wavefront form costs
    // ~2 ns there and grows with h, not h^2 — measured faster at every batch size from 1024-bit keys up (decrypt at 2^13 rows
    // +7.7 %, 2^14 +4.2 %, 2^18 +1.5 % at 2048 bits, profiles/r03p_wave_tail_every_batch_size.txt).  Narrower keys keep it for
    // small batches only: there a wavefront per ciphertext is mostly idle lanes.
    ctx->tail_wave_per_cu = ctx->priv.tail.h >= 16 ? ((size_t)1 << 24) : 16;
    if (const char* d = getenv("PHE_HIP_WAVE_TAIL_PER_CU")) ctx->tail_wave_per_cu = (size_t)std::max(0, atoi(d));
    if (!rc) {
        const size_t lds = (size_t)tail_ws_words(ctx->priv.tail.h) * tail_block(ctx->priv.tail.h) * 4;
        if (lds > 160 * 1024) rc = fail(PHE_HIP_EINVAL, "p/q too wide for the CRT tail kernel");
        else if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)k_decrypt_tail, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) rc = fail(PHE_HIP_EHIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
        }
    }
    if (!rc) rc = setup_owner_encrypt(ctx);
    if (rc) {
        phe_hip_ctx_destroy(ctx);
        return rc;
    }
    ctx->has_private = true;
    *out = ctx;
    return PHE_HIP_OK;
}

void phe_hip_ctx_destroy(phe_hip_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    std::vector<uint32_t*> bufs = {ctx->d_nsplit.blob, ctx->d_psplit.blob, ctx->d_qsplit.blob,
                        ctx->d_nsq.blob, ctx->d_psq.blob, ctx->d_qsq.blob, ctx->d_exp_n.ops, ctx->d_exp_p.ops,
                        ctx->d_exp_q.ops, ctx->d_tail.blob, ctx->table, ctx->table2, ctx->scratch, ctx->partial, ctx->lookup, (uint32_t*)ctx->flags, ctx->stage[0],
                        ctx->stage[1], ctx->stage[2], ctx->owner_blob, ctx->d_nunit.blob, ctx->unit_tmp, ctx->tail_wave_blob, ctx->item_sched,
                        ctx->tmul_blob, ctx->tmul_cols};
    for (const auto& R : ctx->pub_rungs) { bufs.push_back(R.nsq.blob); bufs.push_back(R.nsplit.blob); bufs.push_back(R.nunit.blob); }
    for (const auto& R : ctx->priv_rungs) { bufs.push_back(R.psq.blob); bufs.push_back(R.qsq.blob); bufs.push_back(R.psplit.blob); bufs.push_back(R.qsplit.blob); }
    for (uint32_t* b : bufs)
        if (b) (void)hipFree(b);
    if (ctx->mapped_host) (void)hipHostFree(ctx->mapped_host);
    for (int k = 0; k < 2; ++k) {
        for (int j = 0; j < 3; ++j) {
            if (ctx->pipe.pin[k][j]) (void)hipHostFree(ctx->pipe.pin[k][j]);
            if (ctx->pipe.dev[k][j]) (void)hipFree(ctx->pipe.dev[k][j]);
        }
        hipEvent_t evs[3] = {ctx->pipe.ev_in[k], ctx->pipe.ev_comp[k], ctx->pipe.ev_out[k]};
        for (hipEvent_t e : evs)
            if (e) (void)hipEventDestroy(e);
    }
    hipStream_t pipes[3] = {ctx->pipe.s_in, ctx->pipe.s_comp, ctx->pipe.s_out};
    for (hipStream_t st : pipes)
        if (st) (void)hipStreamDestroy(st);
    if (ctx->ev_busy) (void)hipEventDestroy(ctx->ev_busy);
    delete ctx;
}

int phe_hip_ctx_info(const phe_hip_ctx* ctx, int* n_limbs, int* ct_limbs, int* lane_limbs_pub, int* lane_limbs_priv,
                     int* rows_in_flight, int* has_private) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (n_limbs) *n_limbs = ctx->pub.s1;
    if (ct_limbs) *ct_limbs = ctx->pub.s2;
    // geometry is reported as G*100 + L (e.g. 818 = groups of 8 lanes x 18 limbs of 29 bits)
    // (the geometry of the uniform-exponent kernels in use: split-modulus halves when that engine is on)
    const bool sp_pub = ctx->use_split && ctx->pub.nsplit.G, sp_priv = ctx->use_split && ctx->priv.psplit.G;
    if (lane_limbs_pub) *lane_limbs_pub = sp_pub ? ctx->pub.nsplit.G * 100 + ctx->pub.nsplit.L : ctx->pub.nsq.G * 100 + ctx->pub.nsq.L;
    if (lane_limbs_priv)
        *lane_limbs_priv = !ctx->has_private ? 0 : sp_priv ? ctx->priv.psplit.G * 100 + ctx->priv.psplit.L
                                                           : ctx->priv.psq.G * 100 + ctx->priv.psq.L;
    const int g_pub = sp_pub ? ctx->pub.nsplit.G : std::max(1, ctx->pub.nsq.G);
    if (rows_in_flight) *rows_in_flight = ctx->n_cus * std::max(1, ctx->blocks_per_cu ? ctx->blocks_per_cu : 2) * (kBlock / g_pub);
    if (has_private) *has_private = ctx->has_private ? 1 : 0;
    return PHE_HIP_OK;
}

int phe_hip_ctx_engine(const phe_hip_ctx* ctx, int* split_pub, int* split_priv) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (split_pub) *split_pub = (ctx->use_split && ctx->pub.nsplit.G) ? 1 : 0;
    if (split_priv) *split_priv = (ctx->has_private && ctx->use_split && ctx->priv.psplit.G && ctx->priv.qsplit.G) ? 1 : 0;
    return PHE_HIP_OK;
}

int phe_hip_ctx_set_blocks_per_cu(phe_hip_ctx* ctx, int blocks_per_cu) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (blocks_per_cu < 0 || blocks_per_cu > 8) return fail(PHE_HIP_EINVAL, "blocks_per_cu must be in 0..8 (0 = automatic)");
    ctx->blocks_per_cu = blocks_per_cu;
    return PHE_HIP_OK;
}

int phe_hip_ctx_set_group(phe_hip_ctx* ctx, int group) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (group != 0 && group != 1 && group != 2 && group != 4 && group != 8 && group != 16 && group != 64)
        return fail(PHE_HIP_EINVAL, "group must be 0 (by batch size), 1, 2, 4, 8, 16 or 64");
    ctx->force_group = group;
    return PHE_HIP_OK;
}

int phe_hip_ctx_ladder(const phe_hip_ctx* ctx, int* pub_geoms, int* priv_geoms, int capacity, int* n_pub, int* n_priv) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    const bool sp_pub = ctx->use_split && ctx->pub.nsplit.G, sp_priv = ctx->use_split && ctx->priv.psplit.G;
    int k = 0;
    const auto put = [&](int* dst, int code) { if (dst && k < capacity) dst[k] = code; ++k; };
    put(pub_geoms, sp_pub ? geom_code(ctx->pub.nsplit.G, ctx->pub.nsplit.L) : geom_code(ctx->pub.nsq.G, ctx->pub.nsq.L));
    for (const auto& R : ctx->pub_rungs) put(pub_geoms, sp_pub ? geom_code(R.plan.nsplit.G, R.plan.nsplit.L) : geom_code(R.plan.nsq.G, R.plan.nsq.L));
    if (n_pub) *n_pub = k;
    k = 0;
    if (ctx->has_private) {
        put(priv_geoms, sp_priv ? geom_code(ctx->priv.psplit.G, ctx->priv.psplit.L) : geom_code(ctx->priv.psq.G, ctx->priv.psq.L));
        for (const auto& R : ctx->priv_rungs) put(priv_geoms, sp_priv ? geom_code(R.plan.psplit.G, R.plan.psplit.L) : geom_code(R.plan.psq.G, R.plan.psq.L));
    }
    if (n_priv) *n_priv = k;
    return PHE_HIP_OK;
}

int phe_hip_ctx_load_ladder(phe_hip_ctx* ctx, const char* table, int* accepted) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (accepted) *accepted = 0;
    ctx->ladder.clear();
    if (!table) return PHE_HIP_OK;  // (null: back to the estimate)
    // lines "key_bits family G rows ns" (anything after '#' ignored); only the lines of this context's key width are kept
    std::map<int, std::vector<phe_hip_ctx::LadderPoint>> got;
    int count = 0;
    const char* p = table;
    while (*p) {
        const char* e = strchr(p, '\n');
        std::string line(p, e ? (size_t)(e - p) : strlen(p));
        p = e ? e + 1 : p + line.size();
        const size_t hash = line.find('#');
        if (hash != std::string::npos) {
            // a header comment may say what the table was measured ON ("arch: gfx950", "cus: 256"): launch times of another chip
            // or another CU count must not pick this device's rungs — such a table is not an error, it is simply not taken
            const std::string note = line.substr(hash);
            const size_t at_cus = note.find("cus:"), at_arch = note.find("arch:");
            if (at_cus != std::string::npos) {
                int cus = 0;
                if (sscanf(note.c_str() + at_cus + 4, "%d", &cus) == 1 && cus != ctx->n_cus) return PHE_HIP_OK;
            }
            if (at_arch != std::string::npos) {
                char arch[64] = {0};
                if (sscanf(note.c_str() + at_arch + 5, "%63s", arch) == 1 && ctx->arch.compare(0, strlen(arch), arch) != 0) return PHE_HIP_OK;
            }
            line.resize(hash);
        }
        int bits = 0, family = 0, G = 0;
        double rows = 0, ns = 0;
        if (sscanf(line.c_str(), "%d %d %d %lf %lf", &bits, &family, &G, &rows, &ns) != 5) continue;
        if (bits != 32 * ctx->pub.s1 || family < 0 || family > 3 || G < 1 || G > 64 || !(rows >= 1) || !(ns > 0)) continue;
        got[family * 256 + G].push_back({rows, ns});
        ++count;
    }
    for (auto& kv : got) {
        std::sort(kv.second.begin(), kv.second.end(), [](const phe_hip_ctx::LadderPoint& a, const phe_hip_ctx::LadderPoint& b) { return a.rows < b.rows; });
        for (size_t i = 1; i < kv.second.size(); ++i)
            if (kv.second[i].rows == kv.second[i - 1].rows) return fail(PHE_HIP_EINVAL, "ladder table: a batch size twice for one rung");
    }
    ctx->ladder.swap(got);
    if (accepted) *accepted = count;
    return PHE_HIP_OK;
}

int phe_hip_ctx_last_launch(const phe_hip_ctx* ctx, int* path, int* geom_pub, int* geom_priv) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (path) *path = ctx->last_path;
    if (geom_pub) *geom_pub = ctx->last_geom_pub;
    if (geom_priv) *geom_priv = ctx->last_geom_priv;
    return PHE_HIP_OK;
}

// ---- device-pointer entry points -----------------------------------------------------------------
// the bare power r^n through the scaled modulus (key_setup.h PublicPlan::nunit): throughput geometry only
// r^n modulo the scaled modulus n' = k*n on the rung this batch takes (every rung but the whole-wave one may carry it:
// key_setup.h PublicPlan::nunit); -1: not offered — the plain kernels then.  PHE_HIP_FORCE_UNIT: rung 0's whatever the batch.
static int unit_rung(const phe_hip_ctx* ctx, size_t batch) {
    if (!ctx->use_split) return -1;
    const auto has = [&](int k) {
        const DevSplit& u = k == 0 ? ctx->d_nunit : ctx->pub_rungs[(size_t)k - 1].nunit;
        return u.G != 0 && nsq_rung(ctx, k).G != 0;
    };
    if (ctx->force_unit) return has(0) ? 0 : -1;
    const int k = pick_nsplit_rung(ctx, batch, kFamFixedExp);
    return has(k) ? k : -1;
}
static const DevSplit& nunit_rung(const phe_hip_ctx* ctx, int k) { return k == 0 ? ctx->d_nunit : ctx->pub_rungs[(size_t)k - 1].nunit; }
static int unit_words_rung(const phe_hip_ctx* ctx, int k) { return k == 0 ? ctx->pub.unit_words : ctx->pub_rungs[(size_t)k - 1].plan.unit_words; }
static int unit_power(phe_hip_ctx* ctx, int rung, const uint32_t* r, size_t batch, hipStream_t st) {
    const size_t w = (size_t)unit_words_rung(ctx, rung);
    int rc = ensure_words(&ctx->unit_tmp, &ctx->unit_tmp_words, batch * w);
    if (rc) return rc;
    return launch_split<kModeEncrypt>(ctx, nunit_rung(ctx, rung), ctx->d_exp_n, r, ctx->pub.s1, nullptr, ctx->pub.s1, ctx->unit_tmp,
                                      (int)w, batch, st, false, true);
}

int phe_hip_encrypt_dev(phe_hip_ctx* ctx, const uint32_t* m, const uint32_t* r, uint32_t* c, size_t batch, void* stream) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (batch == 0) return PHE_HIP_OK;
    if (!m || !r || !c) return fail(PHE_HIP_EINVAL, "null buffer");
    if (int rc = bind_device(ctx)) return rc;
    PHE_CTX_ORDER(ctx, stream);
    ctx->last_path = 0;
    if (const DevSplit& sp = pick_nsplit(ctx, batch, kFamFixedExp); !ctx->force_unit && late_offered(ctx, sp) && !ab_offered(ctx, sp, batch, 1)) {
        // a small batch on a rung of 16 lanes or of the whole wave: the late sweeps (scaled modulus, the way out modulo n with the
        // plaintext factor folded in: one launch, no wide scratch rows, no product pass)
        ctx->last_path = kPathLate;
        ctx->last_geom_pub = geom_code(sp.G, sp.q_L);
        return launch_late(ctx, kModeEncrypt, sp, ctx->d_exp_n, r, ctx->pub.s1, m, ctx->pub.s1, c, ctx->pub.s2, batch, (hipStream_t)stream);
    }
    if (const int ur = unit_rung(ctx, batch); ur >= 0 && !(late_offered(ctx, nsplit_rung(ctx, ur)) && !ctx->force_unit)) {
        ctx->last_path = kPathUnit;
        ctx->last_geom_pub = geom_code(nunit_rung(ctx, ur).G, nunit_rung(ctx, ur).L);
        // r^n modulo the scaled modulus n'^2 (no multiply per quotient digit), then ONE pass of the product kernel takes
        // the residue modulo n'^2 to (1 + n*m) * r^n mod n^2 (it accepts any a < R): the same canonical ciphertext
        int rc = unit_power(ctx, ur, r, batch, (hipStream_t)stream);
        if (rc) return rc;
        const size_t s1 = (size_t)ctx->pub.s1, s2 = (size_t)ctx->pub.s2, w = (size_t)unit_words_rung(ctx, ur);
        return launch_mul(ctx, nsq_rung(ctx, ur), ctx->unit_tmp, w, m, s1, c, s2, ctx->pub.s2, batch, (hipStream_t)stream, ctx->pub.s1, 0,
                          (int)w);
    }
    if (const DevSplit& sp = pick_nsplit(ctx, batch, kFamFixedExp); ctx->use_split && sp.G) {
        ctx->last_geom_pub = geom_code(sp.G, sp.L);
        if (ab_offered(ctx, sp, batch, 1)) {
            ctx->last_path = kPathWavePairs;
            return launch_split_ab(ctx, kModeEncrypt, sp, ctx->d_exp_n, nullptr, nullptr, r, ctx->pub.s1, m, ctx->pub.s1, c, nullptr,
                                   ctx->pub.s2, batch, (hipStream_t)stream);
        }
        return launch_split<kModeEncrypt>(ctx, sp, ctx->d_exp_n, r, ctx->pub.s1, m, ctx->pub.s1, c, ctx->pub.s2, batch,
                                          (hipStream_t)stream);
    }
    const DevModulus& fw = pick_nsq(ctx, batch);
    ctx->last_geom_pub = geom_code(fw.G, fw.L);
    return launch_uniform<kModeEncrypt>(ctx, fw, ctx->d_exp_n, r, ctx->pub.s1, m, ctx->pub.s1, c, ctx->pub.s2, batch,
                                        (hipStream_t)stream);
}

// raw_encrypt by the holder of the private key: r^n mod n^2 from r^n mod p^2 and r^n mod q^2 (the half-exponentiation
// kernels with the exponent n), lifted by k_crt_lift, times 1 + n*m by the product kernel.  Same bits as phe_hip_encrypt_dev.
// The two half-exponentiations base^e_p mod p^2 -> xp, base^e_q mod q^2 -> xq (rows of S words) of a decrypt (e = p - 1, q - 1)
// or of the key owner's encryption (e = n for both): the rung of the ladder for this batch, the two halves as one grid while
// they fit one residency together, one number on a pair of wavefronts for a handful of numbers.  Sets last_geom_priv and the
// path bits it took (|= into ctx->last_path).
static int launch_crt_halves(phe_hip_ctx* ctx, const DevSchedule& Ep, const DevSchedule& Eq, const uint32_t* base, int base_limbs,
                             uint32_t* xp, uint32_t* xq, int S, size_t batch, hipStream_t st, int family = kFamHalves) {
    int rc = PHE_HIP_OK;
    // the rung of the halves: the two exponentiations are independent, so a batch that cannot fill the chip with one of them
    // runs both side by side (one grid, the q half with its own window tables) and needs only half the lanes
    const int n_rungs = 1 + (int)ctx->priv_rungs.size();
    const auto psplit_of = [&](int k) -> const DevSplit& { return k == 0 ? ctx->d_psplit : ctx->priv_rungs[(size_t)k - 1].psplit; };
    const auto qsplit_of = [&](int k) -> const DevSplit& { return k == 0 ? ctx->d_qsplit : ctx->priv_rungs[(size_t)k - 1].qsplit; };
    const auto psq_of = [&](int k) -> const DevModulus& { return k == 0 ? ctx->d_psq : ctx->priv_rungs[(size_t)k - 1].psq; };
    const auto qsq_of = [&](int k) -> const DevModulus& { return k == 0 ? ctx->d_qsq : ctx->priv_rungs[(size_t)k - 1].qsq; };
    const bool split_ok = ctx->use_split && ctx->d_psplit.G && ctx->d_qsplit.G;
    const auto late_rung = [&](int k) {
        return split_ok && late_offered(ctx, psplit_of(k)) && late_offered(ctx, qsplit_of(k)) && psplit_of(k).q_L == qsplit_of(k).q_L;
    };
    const auto shape = [&](int k) {
        if (split_ok) return split_shape(qsplit_of(k), family, late_rung(k));
        RungShape sh;
        sh.G = qsq_of(k).G;
        sh.L = qsq_of(k).L;
        sh.rows = qsq_of(k).S;
        return sh;
    };
    int rung = pick_rung(ctx, batch, n_rungs, shape, split_ok ? 2 : 1);
    if (split_ok && !(psplit_of(rung).G && qsplit_of(rung).G)) rung = 0;
    const DevSplit& sp_p = psplit_of(rung);
    const DevSplit& sp_q = qsplit_of(rung);
    // side by side (one grid, k_modexp_split_halves) while both halves together fit one residency of workgroups; beyond that
    // each half fills the GPU on its own and they run one after the other
    const bool late = late_rung(rung);
    int occ = ctx->blocks_per_cu;
    if (occ == 0 && split_ok)
        occ = late ? occ_small_rung(sp_p, kModeHalfDecrypt) : PHE_SPLIT_BY_GROUP(sp_p.G, occ_split(sp_p.L, kModeHalfDecrypt));
    const size_t wg_per_half = (batch + (size_t)(kBlock / std::max(1, sp_p.G)) - 1) / (size_t)(kBlock / std::max(1, sp_p.G));
    const bool side_by_side = split_ok && sp_p.G == sp_q.G && sp_p.L == sp_q.L &&
                              2 * wg_per_half <= (size_t)ctx->n_cus * (size_t)std::max(1, occ);
    ctx->last_geom_priv = split_ok ? geom_code(sp_p.G, late ? sp_p.q_L : sp_p.L) : geom_code(psq_of(rung).G, psq_of(rung).L);
    ctx->last_path |= side_by_side ? kPathSideBySide : 0;
    if (split_ok && sp_p.G == sp_q.G && sp_p.L == sp_q.L && ab_offered(ctx, sp_p, batch, 2)) {
        // a handful of ciphertexts: every half-exponentiation on a PAIR of wavefronts (about half the time per product)
        ctx->last_path |= kPathSideBySide | kPathWavePairs;
        rc = launch_split_ab(ctx, kModeHalfDecrypt, sp_p, Ep, &sp_q, &Eq, base, base_limbs, nullptr, 0, xp, xq, S, batch, st);
        if (rc) return rc;
    } else if (late) {
        // the small-batch rungs on the late sweeps: one grid for both halves while they fit one residency together
        ctx->last_path |= kPathLate;
        if (side_by_side) {
            rc = launch_late_halves(ctx, sp_p, Ep, sp_q, Eq, base, base_limbs, xp, xq, S, batch, st);
        } else {
            rc = launch_late(ctx, kModeHalfDecrypt, sp_p, Ep, base, base_limbs, nullptr, 0, xp, S, batch, st);
            if (!rc) rc = launch_late(ctx, kModeHalfDecrypt, sp_q, Eq, base, base_limbs, nullptr, 0, xq, S, batch, st);
        }
        if (rc) return rc;
    } else if (side_by_side) {
        rc = launch_split_halves(ctx, sp_p, Ep, sp_q, Eq, base, base_limbs, xp, xq, S, batch, st);
        if (rc) return rc;
    } else {
        if (split_ok)
            rc = launch_split<kModeHalfDecrypt>(ctx, sp_p, Ep, base, base_limbs, nullptr, 0, xp, S, batch, st);
        else
            rc = launch_uniform<kModeHalfDecrypt>(ctx, psq_of(rung), Ep, base, base_limbs, nullptr, 0, xp, S, batch, st);
        if (rc) return rc;
        if (split_ok)
            rc = launch_split<kModeHalfDecrypt>(ctx, sp_q, Eq, base, base_limbs, nullptr, 0, xq, S, batch, st);
        else
            rc = launch_uniform<kModeHalfDecrypt>(ctx, qsq_of(rung), Eq, base, base_limbs, nullptr, 0, xq, S, batch, st);
        if (rc) return rc;
    }
    return PHE_HIP_OK;
}

int phe_hip_encrypt_owner_dev(phe_hip_ctx* ctx, const uint32_t* m, const uint32_t* r, uint32_t* c, size_t batch, void* stream) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (!ctx->has_private) return fail(PHE_HIP_EINVAL, "owner encryption needs a private-key context");
    if (!ctx->owner_blob) return fail(PHE_HIP_EINVAL, "owner encryption is not offered for this key width");
    if (batch == 0) return PHE_HIP_OK;
    if (!m || !r || !c) return fail(PHE_HIP_EINVAL, "null buffer");
    if (int rc = bind_device(ctx)) return rc;
    PHE_CTX_ORDER(ctx, stream);
    hipStream_t st = (hipStream_t)stream;
    ctx->last_path = kPathOwner;
    const int S = (std::max(ctx->priv.psq.bits, ctx->priv.qsq.bits) + 31) / 32;
    int rc = ensure_words(&ctx->scratch, &ctx->scratch_words, (size_t)2 * batch * S);
    if (rc) return rc;
    uint32_t* yp = ctx->scratch;
    uint32_t* yq = ctx->scratch + batch * (size_t)S;
    // the halves on the rung this batch size calls for (the lift below takes canonical residues: any rung's will do)
    // (its own family: the measured ladder's family 2 was timed with the decrypt's exponents p - 1, q - 1; the exponent n here is
    //  twice as long, so this job keeps the estimate)
    rc = launch_crt_halves(ctx, ctx->d_exp_n, ctx->d_exp_n, r, ctx->pub.s1, yp, yq, S, batch, st, kFamHalvesLongExp);
    if (rc) return rc;
    const int QS = ctx->d_qsq.S;
    CrtLiftArgs A;
    A.mod = ctx->d_qsq.c;
    A.kr = ctx->owner_blob;
    A.nkr = ctx->owner_blob + QS;
    A.psq = ctx->owner_blob + 2 * QS;
    A.yp = yp;
    A.yq = yq;
    A.x_stride = (size_t)S;
    A.x_limbs = S;
    A.out = c;
    A.out_limbs = ctx->pub.s2;
    A.batch = batch;
    const int blocks = grid_blocks(ctx, batch, ctx->owner_G, 2);
    if (PHE_SPLIT_BY_GROUP(ctx->owner_G, launch_crt_lift(ctx->owner_L, blocks, st, A)) < 0)
        return fail(PHE_HIP_EINVAL, "unsupported geometry for the CRT lift");
    HIP_TRY(hipGetLastError());
    // c <- r^n * (1 + n*m) mod n^2, in place (every limb group reads its row before it writes it)
    const size_t s1 = (size_t)ctx->pub.s1, s2 = (size_t)ctx->pub.s2;
    return launch_mul(ctx, ctx->d_nsq, c, s2, m, s1, c, s2, ctx->pub.s2, batch, st, ctx->pub.s1);
}

int phe_hip_ctx_owner_encrypt(const phe_hip_ctx* ctx, int* offered) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (!offered) return fail(PHE_HIP_EINVAL, "null pointer");
    *offered = (ctx->has_private && ctx->owner_blob) ? 1 : 0;
    return PHE_HIP_OK;
}

int phe_hip_obfuscate_dev(phe_hip_ctx* ctx, const uint32_t* c_in, const uint32_t* r, uint32_t* c_out, size_t batch,
                          void* stream) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (batch == 0) return PHE_HIP_OK;
    if (!c_in || !r || !c_out) return fail(PHE_HIP_EINVAL, "null buffer");
    if (int rc = bind_device(ctx)) return rc;
    PHE_CTX_ORDER(ctx, stream);
    ctx->last_path = 0;
    if (const DevSplit& sp = pick_nsplit(ctx, batch, kFamFixedExp);
        c_in != c_out && !ctx->force_unit && late_offered(ctx, sp) && !ab_offered(ctx, sp, batch, 1) && !getenv("PHE_HIP_FUSED_OBFUSCATE")) {
        // r^n (the bare power) by the late sweeps, then the product with the ciphertext
        ctx->last_path = kPathLate;
        ctx->last_geom_pub = geom_code(sp.G, sp.q_L);
        int rc = launch_late(ctx, kModeEncrypt, sp, ctx->d_exp_n, r, ctx->pub.s1, nullptr, 0, c_out, ctx->pub.s2, batch, (hipStream_t)stream);
        if (rc) return rc;
        const size_t s2 = (size_t)ctx->pub.s2;
        return launch_mul(ctx, pick_nsq(ctx, batch), c_out, s2, c_in, s2, c_out, s2, ctx->pub.s2, batch, (hipStream_t)stream);
    }
    if (const int ur = unit_rung(ctx, batch); ur >= 0 && !getenv("PHE_HIP_FUSED_OBFUSCATE") &&
                                                !(late_offered(ctx, nsplit_rung(ctx, ur)) && !ctx->force_unit && c_in != c_out)) {
        ctx->last_path = kPathUnit;
        ctx->last_geom_pub = geom_code(nunit_rung(ctx, ur).G, nunit_rung(ctx, ur).L);
        // r^n modulo the scaled modulus, then the product with the ciphertext brings it to n^2 (in-place calls included:
        // the power sits in its own buffer)
        int rc = unit_power(ctx, ur, r, batch, (hipStream_t)stream);
        if (rc) return rc;
        const size_t s2 = (size_t)ctx->pub.s2, w = (size_t)unit_words_rung(ctx, ur);
        return launch_mul(ctx, nsq_rung(ctx, ur), ctx->unit_tmp, w, c_in, s2, c_out, s2, ctx->pub.s2, batch, (hipStream_t)stream, 0, 0,
                          (int)w);
    }
    if (const DevSplit& sp = pick_nsplit(ctx, batch, kFamFixedExp); ctx->use_split && sp.G) {
        ctx->last_geom_pub = geom_code(sp.G, sp.L);
        if (c_in != c_out && ab_offered(ctx, sp, batch, 1)) {
            // a handful of ciphertexts: r^n of each on a pair of wavefronts (the bare power: no plaintext factor), then the product
            ctx->last_path = kPathWavePairs;
            int rc = launch_split_ab(ctx, kModeEncrypt, sp, ctx->d_exp_n, nullptr, nullptr, r, ctx->pub.s1, nullptr, 0, c_out, nullptr,
                                     ctx->pub.s2, batch, (hipStream_t)stream);
            if (rc) return rc;
            const size_t s2 = (size_t)ctx->pub.s2;
            return launch_mul(ctx, pick_nsq(ctx, batch), c_out, s2, c_in, s2, c_out, s2, ctx->pub.s2, batch, (hipStream_t)stream);
        }
        if (c_in != c_out && !getenv("PHE_HIP_FUSED_OBFUSCATE")) {
            // r^n with the encrypt instantiation (no plaintext factor), then one k_mulmod by the ciphertext: the fused
            // kModeObfuscate instantiation spills more (PMC: 94 KB written per element against 20 KB,
            // profiles/r01p_rocprofv3_pmc_ops.txt) and measured 559.7 k/s against 566.6 k/s for this form on the same box
            int rc = launch_split<kModeEncrypt>(ctx, sp, ctx->d_exp_n, r, ctx->pub.s1, nullptr, ctx->pub.s1, c_out, ctx->pub.s2,
                                                batch, (hipStream_t)stream);
            if (rc) return rc;
            const size_t s2 = (size_t)ctx->pub.s2;
            return launch_mul(ctx, pick_nsq(ctx, batch), c_out, s2, c_in, s2, c_out, s2, ctx->pub.s2, batch, (hipStream_t)stream);
        }
        ctx->last_path = kPathFusedObfuscate;
        return launch_split<kModeObfuscate>(ctx, sp, ctx->d_exp_n, r, ctx->pub.s1, c_in, ctx->pub.s2, c_out, ctx->pub.s2,
                                            batch, (hipStream_t)stream);
    }
    return launch_uniform<kModeObfuscate>(ctx, pick_nsq(ctx, batch), ctx->d_exp_n, r, ctx->pub.s1, c_in, ctx->pub.s2, c_out,
                                          ctx->pub.s2, batch, (hipStream_t)stream);
}

int phe_hip_decrypt_dev(phe_hip_ctx* ctx, const uint32_t* c, uint32_t* m, size_t batch, void* stream) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (!ctx->has_private) return fail(PHE_HIP_EINVAL, "decrypt needs a private-key context");
    if (batch == 0) return PHE_HIP_OK;
    if (!c || !m) return fail(PHE_HIP_EINVAL, "null buffer");
    if (int rc = bind_device(ctx)) return rc;
    PHE_CTX_ORDER(ctx, stream);
    // x_p, x_q rows in 32-bit words; only their low h words are read by the tail
    const int S = (std::max(ctx->priv.psq.bits, ctx->priv.qsq.bits) + 31) / 32;
    int rc = ensure_words(&ctx->scratch, &ctx->scratch_words, (size_t)2 * batch * S);
    if (rc) return rc;
    uint32_t* xp = ctx->scratch;
    uint32_t* xq = ctx->scratch + batch * (size_t)S;
    hipStream_t st = (hipStream_t)stream;
    ctx->last_path = 0;
    rc = launch_crt_halves(ctx, ctx->d_exp_p, ctx->d_exp_q, c, ctx->pub.s2, xp, xq, S, batch, st);
    if (rc) return rc;
    // the tail: one ciphertext per wavefront on the whole-wave sweeps (same bits as the per-thread kernel below, which is a
    // 0.4 ms serial chain through LDS at 2048-bit keys and serves narrow keys' large batches and PHE_HIP_NO_WAVE_TAIL)
    if (ctx->tail_wave_L && batch <= (size_t)ctx->n_cus * ctx->tail_wave_per_cu && batch <= (size_t)0x7fffffff) {
        TailWaveArgs W;
        W.k = ctx->d_tail_wave;
        W.xp = xp;
        W.xq = xq;
        W.x_stride = S;
        W.m_out = m;
        W.out_limbs = ctx->pub.s1;
        W.batch = batch;
        if (PHE_SPLIT_BY_GROUP(64, launch_tail_wave(ctx->tail_wave_L, (int)batch, st, W)) < 0)
            return fail(PHE_HIP_EINVAL, "no wave tail kernel for this width");
        HIP_TRY(hipGetLastError());
        ctx->last_path |= kPathWaveTail;
        return PHE_HIP_OK;
    }
    TailArgs T;
    T.k = ctx->d_tail.k;
    T.xp = xp;
    T.xq = xq;
    T.x_stride = S;
    T.m_out = m;
    T.out_limbs = ctx->pub.s1;
    T.batch = batch;
    const int tb = tail_block(T.k.h);
    const size_t lds = (size_t)tail_ws_words(T.k.h) * tb * 4;
    const int blocks = (int)((batch + tb - 1) / tb);
    k_decrypt_tail<<<dim3(blocks), dim3(tb), lds, st>>>(T);
    HIP_TRY(hipGetLastError());
    return PHE_HIP_OK;
}

int phe_hip_mulmod_dev(phe_hip_ctx* ctx, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t batch, void* stream) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (batch == 0) return PHE_HIP_OK;
    if (!a || !b || !out) return fail(PHE_HIP_EINVAL, "null buffer");
    if (int rc = bind_device(ctx)) return rc;
    PHE_CTX_ORDER(ctx, stream);
    const size_t s2 = (size_t)ctx->pub.s2;
    ctx->last_path = 0;
    return launch_mul(ctx, pick_nsq(ctx, batch), a, s2, b, s2, out, s2, ctx->pub.s2, batch, (hipStream_t)stream, 0, 0, 0, true);
}

int phe_hip_montmul_dev(phe_hip_ctx* ctx, const uint32_t* a, const uint32_t* b, int b_is_row, uint32_t* out, size_t batch,
                        void* stream) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (batch == 0) return PHE_HIP_OK;
    if (!a || !b || !out) return fail(PHE_HIP_EINVAL, "null buffer");
    if (int rc = bind_device(ctx)) return rc;
    PHE_CTX_ORDER(ctx, stream);
    const size_t s2 = (size_t)ctx->pub.s2;
    if (ctx->d_nsq.G == 0) return fail(PHE_HIP_EINVAL, "one-product form not available for this key width");
    // always the throughput geometry: R must not depend on the batch size
    return launch_mul(ctx, ctx->d_nsq, a, s2, b, b_is_row ? 0 : s2, out, s2, ctx->pub.s2, batch, (hipStream_t)stream, 0, 1);
}

int phe_hip_mont_radix_bits(phe_hip_ctx* ctx, int* bits) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (!bits) return fail(PHE_HIP_EINVAL, "null pointer");
    if (ctx->d_nsq.G == 0) return fail(PHE_HIP_EINVAL, "one-product form not available for this key width");
    *bits = phe::kRadixBits * ctx->d_nsq.S;
    return PHE_HIP_OK;
}

int phe_hip_add_plain_dev(phe_hip_ctx* ctx, const uint32_t* c, const uint32_t* m, uint32_t* out, size_t batch, void* stream) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (batch == 0) return PHE_HIP_OK;
    if (!c || !m || !out) return fail(PHE_HIP_EINVAL, "null buffer");
    if (int rc = bind_device(ctx)) return rc;
    PHE_CTX_ORDER(ctx, stream);
    const size_t s1 = (size_t)ctx->pub.s1, s2 = (size_t)ctx->pub.s2;
    return launch_mul(ctx, pick_nsq(ctx, batch), c, s2, m, s1, out, s2, ctx->pub.s2, batch, (hipStream_t)stream, ctx->pub.s1);
}

// ---- resident rows in the pair form (split_core.h "resident ciphertext rows") ----------------------------------------------
// The rows are H | H limbs of rung 0's pair geometry; a wider rung with the same H can serve a small batch of them.
static const DevSplit& pick_pair_split(const phe_hip_ctx* ctx, size_t batch) {
    const int n_rungs = 1 + (int)ctx->pub_rungs.size();
    const int k = pick_rung(ctx, batch, n_rungs, [&](int r) {
        const DevSplit& sp = nsplit_rung(ctx, r);
        return (sp.H == ctx->d_nsplit.H && sp.rows == sp.H) ? split_shape(sp) : RungShape();
    });
    const DevSplit& sp = nsplit_rung(ctx, k);
    return sp.H == ctx->d_nsplit.H ? sp : ctx->d_nsplit;
}
static int pair_launch(phe_hip_ctx* ctx, int op, const uint32_t* a, const uint32_t* b, size_t b_stride, int b_limbs, uint32_t* out,
                       size_t batch, hipStream_t st) {
    if (!(ctx->use_split && ctx->d_nsplit.G)) return fail(PHE_HIP_EINVAL, "the pair form needs the split-modulus engine");
    const DevSplit& sp = pick_pair_split(ctx, batch);
    PairArgs A;
    A.mod = sp.c;
    A.a = a;
    A.b = b;
    A.out = out;
    A.b_stride = b_stride;
    A.limbs = ctx->pub.s2;
    A.chunks = chunks_for(ctx->pub.s2, sp.rows);
    A.b_limbs = b_limbs;
    A.batch = batch;
    ctx->last_path = 0;  // (phe_hip_ctx_last_launch describes THIS call: no path bit of an earlier encrypt / decrypt survives)
    ctx->last_geom_pub = geom_code(sp.G, sp.L);
    const int blocks = grid_blocks(ctx, batch, sp.G, 2);
    if (PHE_SPLIT_BY_GROUP(sp.G, launch_pair(sp.L, op, blocks, st, A)) < 0) return fail(PHE_HIP_EINVAL, "unsupported split geometry");
    HIP_TRY(hipGetLastError());
    return PHE_HIP_OK;
}

int phe_hip_pair_words(const phe_hip_ctx* ctx, int* words) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (!words) return fail(PHE_HIP_EINVAL, "null pointer");
    if (!(ctx->use_split && ctx->d_nsplit.G)) return fail(PHE_HIP_EINVAL, "the pair form needs the split-modulus engine");
    *words = 2 * ctx->d_nsplit.H;
    return PHE_HIP_OK;
}

int phe_hip_to_pair_dev(phe_hip_ctx* ctx, const uint32_t* c, uint32_t* pair, size_t batch, void* stream) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (batch == 0) return PHE_HIP_OK;
    if (!c || !pair) return fail(PHE_HIP_EINVAL, "null buffer");
    if (int rc = bind_device(ctx)) return rc;
    PHE_CTX_ORDER(ctx, stream);
    return pair_launch(ctx, 0, c, nullptr, 0, 0, pair, batch, (hipStream_t)stream);
}

int phe_hip_from_pair_dev(phe_hip_ctx* ctx, const uint32_t* pair, const uint32_t* m, uint32_t* c, size_t batch, void* stream) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (batch == 0) return PHE_HIP_OK;
    if (!c || !pair) return fail(PHE_HIP_EINVAL, "null buffer");
    if (int rc = bind_device(ctx)) return rc;
    PHE_CTX_ORDER(ctx, stream);
    return pair_launch(ctx, 1, pair, m, 0, ctx->pub.s1, c, batch, (hipStream_t)stream);
}

int phe_hip_pair_mul_dev(phe_hip_ctx* ctx, const uint32_t* a, const uint32_t* b, int b_is_row, uint32_t* out, size_t batch,
                         void* stream) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (batch == 0) return PHE_HIP_OK;
    if (!a || !b || !out) return fail(PHE_HIP_EINVAL, "null buffer");
    if (int rc = bind_device(ctx)) return rc;
    PHE_CTX_ORDER(ctx, stream);
    return pair_launch(ctx, 2, a, b, b_is_row ? 0 : (size_t)2 * ctx->d_nsplit.H, 0, out, batch, (hipStream_t)stream);
}

// out[i] = a[i]^e[i] on rows in the pair form (both): _raw_mul of resident vectors (phe/paillier.py:751) without the conversion
// into the pair form and the exit from it that phe_hip_powmod_dev pays per element
int phe_hip_pair_powmod_dev(phe_hip_ctx* ctx, const uint32_t* a, const uint32_t* e, int exp_limbs, int max_exp_bits, uint32_t* out,
                            size_t batch, void* stream) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (batch == 0) return PHE_HIP_OK;
    if (!a || !e || !out || exp_limbs < 1) return fail(PHE_HIP_EINVAL, "null buffer / exp_limbs");
    if (max_exp_bits <= 0 || max_exp_bits > 32 * exp_limbs) max_exp_bits = 32 * exp_limbs;
    if (int rc = bind_device(ctx)) return rc;
    if (!(ctx->use_split && ctx->d_nsplit.G)) return fail(PHE_HIP_EINVAL, "the pair form needs the split-modulus engine");
    PHE_CTX_ORDER(ctx, stream);
    const DevSplit& sp = pick_pair_split(ctx, batch);
    ctx->last_path = 0;
    ctx->last_geom_pub = geom_code(sp.G, sp.L);
    return launch_var_split(ctx, sp, a, 0, e, exp_limbs, max_exp_bits, out, 0, batch, (hipStream_t)stream, true);
}

// out = the product of all `batch` pair rows (one pair row): the pairwise tree of EncryptedVector.sum() — sum(enc_list) in the
// reference is a left-to-right chain of _raw_add (phe/paillier.py:705-719); the product of residues does not depend on the
// order — as log2(batch) launches queued back to back, no host round trip between the levels.
int phe_hip_pair_reduce_dev(phe_hip_ctx* ctx, const uint32_t* pair, size_t batch, uint32_t* out, void* stream) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (!pair || !out || batch == 0) return fail(PHE_HIP_EINVAL, "null buffer / empty vector");
    if (int rc = bind_device(ctx)) return rc;
    if (!(ctx->use_split && ctx->d_nsplit.G)) return fail(PHE_HIP_EINVAL, "the pair form needs the split-modulus engine");
    PHE_CTX_ORDER(ctx, stream);
    hipStream_t st = (hipStream_t)stream;
    const size_t w = (size_t)2 * ctx->d_nsplit.H;
    if (batch == 1) {
        HIP_TRY(hipMemcpyAsync(out, pair, w * 4, hipMemcpyDeviceToDevice, st));
        return PHE_HIP_OK;
    }
    size_t cur = batch / 2;  // rows after the first level (+ the unpaired row, if any)
    int rc = ensure_words(&ctx->partial, &ctx->partial_words, (cur + 1) * w);
    if (rc) return rc;
    uint32_t* P = ctx->partial;
    rc = pair_launch(ctx, 2, pair, pair + cur * w, w, 0, P, cur, st);
    if (rc) return rc;
    if (batch & 1) {
        HIP_TRY(hipMemcpyAsync(P + cur * w, pair + (batch - 1) * w, w * 4, hipMemcpyDeviceToDevice, st));
        ++cur;
    }
    while (cur > 1) {
        const size_t half = cur / 2;
        // in place: every limb group reads its two rows before it writes row i, and row i is nobody else's operand
        rc = pair_launch(ctx, 2, P, P + half * w, w, 0, P, half, st);
        if (rc) return rc;
        if (cur & 1) {
            HIP_TRY(hipMemcpyAsync(P + half * w, P + (cur - 1) * w, w * 4, hipMemcpyDeviceToDevice, st));
            cur = half + 1;
        } else {
            cur = half;
        }
    }
    HIP_TRY(hipMemcpyAsync(out, P, w * 4, hipMemcpyDeviceToDevice, st));
    return PHE_HIP_OK;
}

int phe_hip_powmod_dev(phe_hip_ctx* ctx, const uint32_t* base, const uint32_t* e, int exp_limbs, int max_exp_bits,
                       uint32_t* out, size_t batch, void* stream) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (batch == 0) return PHE_HIP_OK;
    if (!base || !e || !out || exp_limbs < 1) return fail(PHE_HIP_EINVAL, "null buffer / exp_limbs");
    if (max_exp_bits <= 0 || max_exp_bits > 32 * exp_limbs) max_exp_bits = 32 * exp_limbs;
    if (int rc = bind_device(ctx)) return rc;
    PHE_CTX_ORDER(ctx, stream);
    ctx->last_path = 0;  // (phe_hip_ctx_last_launch describes THIS call)
    if (const DevSplit& sp = pick_nsplit(ctx, batch, kFamVarExp); ctx->use_split && sp.G) {
        ctx->last_geom_pub = geom_code(sp.G, sp.L);
        return launch_var_split(ctx, sp, base, ctx->pub.s2, e, exp_limbs, max_exp_bits, out, ctx->pub.s2, batch,
                                (hipStream_t)stream);
    }
    const DevModulus& fw = pick_nsq(ctx, batch);
    ctx->last_geom_pub = geom_code(fw.G, fw.L);
    return launch_var(ctx, fw, base, ctx->pub.s2, e, exp_limbs, max_exp_bits, out, ctx->pub.s2, batch, (hipStream_t)stream);
}

// out[r] = prod_i b_i^e[r][i] mod n^2, b_i = base_i or base_inv_i where neg[r][i] — `rows` encrypted dot products over
// the same ciphertexts (rows = 1: phe_hip_multiexp).  Tasks (chunk of the batch, block of rows) go through
// k_multiexp_split — tables built once per task, one shared square-and-multiply ladder per row — and the chunk products
// of every row are joined in place by a pairwise k_mulmod tree over the chunk index.  Without a split geometry (or
// PHE_HIP_ENGINE=full) a single row without negative entries takes the powmod kernel (chunk = 1) and the same tree.
static const DevSplit& pick_pair_split(const phe_hip_ctx* ctx, size_t batch);
static int multiexp_rows_impl(phe_hip_ctx* ctx, const uint32_t* base, const uint32_t* base_inv, const uint32_t* e,
                              const uint8_t* neg, int exp_limbs, int max_exp_bits, uint32_t* out, size_t batch, size_t rows,
                              hipStream_t st, bool pair_in = false) {
    const size_t s2 = (size_t)ctx->pub.s2;
    if (rows == 0) return PHE_HIP_OK;
    if (batch == 0) {  // empty products
        std::vector<uint32_t> ones(rows * s2, 0u);
        for (size_t r = 0; r < rows; ++r) ones[r * s2] = 1u;
        HIP_TRY(hipMemcpyAsync(out, ones.data(), rows * s2 * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        return PHE_HIP_OK;
    }
    if (!base || !e || exp_limbs < 1) return fail(PHE_HIP_EINVAL, "null buffer / exp_limbs");
    if (neg && !base_inv) return fail(PHE_HIP_EINVAL, "a sign mask needs the inverted bases");
    if (max_exp_bits <= 0 || max_exp_bits > 32 * exp_limbs) max_exp_bits = 32 * exp_limbs;
    const bool split = ctx->use_split && ctx->d_nsplit.G;
    if (!split && (rows != 1 || neg))
        return fail(PHE_HIP_EINVAL, "the matrix form needs the split-modulus engine (call row by row on this key)");
    size_t chunk = 1, row_block = 1;
    if (split) {
        // fill the resident groups of the throughput geometry first, then let chunks grow (a group's tables are
        // chunk * (2^w - 1) pairs) and rows share a task's tables
        int per_cu = ctx->blocks_per_cu;
        if (per_cu == 0) per_cu = PHE_SPLIT_BY_GROUP(ctx->d_nsplit.G, occ_multi_split(ctx->d_nsplit.L));
        const size_t resident = (size_t)ctx->n_cus * (size_t)std::max(1, per_cu) * (size_t)(kBlock / ctx->d_nsplit.G);
        // (cap 32 once the batch offers two full rounds of 32-element chunks: +3.6 % measured at 2^20 elements)
        const size_t cap = (batch * rows >= 64 * resident) ? 32 : 16;
        chunk = std::min<size_t>(std::min<size_t>(cap, batch), std::max<size_t>(1, batch * rows / resident));
        if (const char* ev = getenv("PHE_HIP_MULTI_CHUNK")) {
            const int v = atoi(ev);
            if (v >= 1 && v <= 64) chunk = (size_t)v;
        }
        const size_t n_chunks = (batch + chunk - 1) / chunk;
        row_block = std::min<size_t>(std::min<size_t>(64, rows), std::max<size_t>(1, n_chunks * rows / (2 * resident)));
        if (const char* ev = getenv("PHE_HIP_MULTI_ROWBLOCK")) {
            const int v = atoi(ev);
            if (v >= 1 && v <= 1024) row_block = (size_t)v;
        }
    }
    size_t cur = (batch + chunk - 1) / chunk;
    int rc = ensure_words(&ctx->partial, &ctx->partial_words, cur * rows * s2);
    if (rc) return rc;
    uint32_t* P = ctx->partial;
    if (split) {
        const size_t tasks = cur * ((rows + row_block - 1) / row_block);
        // pair_in: the rows are 2H limbs of rung 0's pair geometry — a wider rung with the same H may serve a small job
        const DevSplit& rung = pair_in ? pick_pair_split(ctx, tasks) : pick_nsplit(ctx, tasks);
        rc = launch_multi_split(ctx, rung, base, base_inv, pair_in ? 2 * rung.H : ctx->pub.s2, e, neg, exp_limbs, max_exp_bits,
                                (int)chunk, (int)row_block, P, ctx->pub.s2, batch, rows, st, pair_in);
    } else {
        rc = launch_var(ctx, pick_nsq(ctx, batch), base, ctx->pub.s2, e, exp_limbs, max_exp_bits, P, ctx->pub.s2, batch, st);
    }
    if (rc) return rc;
    const size_t block = rows * s2;  // words of one chunk index
    while (cur > 1) {
        const size_t half = cur / 2;
        rc = launch_mul(ctx, pick_nsq(ctx, half * rows), P, s2, P + half * block, s2, P, s2, ctx->pub.s2, half * rows, st);
        if (rc) return rc;
        if (cur & 1) {  // the unpaired last chunk joins the next level
            HIP_TRY(hipMemcpyAsync(P + half * block, P + 2 * half * block, block * 4, hipMemcpyDeviceToDevice, st));
            cur = half + 1;
        } else {
            cur = half;
        }
    }
    HIP_TRY(hipMemcpyAsync(out, P, block * 4, hipMemcpyDeviceToDevice, st));
    return PHE_HIP_OK;
}

int phe_hip_multiexp_dev(phe_hip_ctx* ctx, const uint32_t* base, const uint32_t* e, int exp_limbs, int max_exp_bits,
                         uint32_t* out, size_t batch, void* stream) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (!out) return fail(PHE_HIP_EINVAL, "null buffer");
    if (int rc = bind_device(ctx)) return rc;
    PHE_CTX_ORDER(ctx, stream);
    return multiexp_rows_impl(ctx, base, nullptr, e, nullptr, exp_limbs, max_exp_bits, out, batch, 1, (hipStream_t)stream);
}

// the same on resident rows in the pair form (phe_hip_to_pair_dev): no conversion into the form per ciphertext; no negative
// entries (those take invert(c), which wants residues).  out: canonical residues, as ever.
int phe_hip_pair_multiexp_rows_dev(phe_hip_ctx* ctx, const uint32_t* pair_base, const uint32_t* e, int exp_limbs, int max_exp_bits,
                                   uint32_t* out, size_t batch, size_t rows, void* stream) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (!out) return fail(PHE_HIP_EINVAL, "null buffer");
    if (!(ctx->use_split && ctx->d_nsplit.G)) return fail(PHE_HIP_EINVAL, "the pair form needs the split-modulus engine");
    if (int rc = bind_device(ctx)) return rc;
    PHE_CTX_ORDER(ctx, stream);
    ctx->last_path = 0;
    return multiexp_rows_impl(ctx, pair_base, nullptr, e, nullptr, exp_limbs, max_exp_bits, out, batch, rows, (hipStream_t)stream, true);
}

int phe_hip_multiexp_rows_dev(phe_hip_ctx* ctx, const uint32_t* base, const uint32_t* base_inv, const uint32_t* e,
                              const uint8_t* neg, int exp_limbs, int max_exp_bits, uint32_t* out, size_t batch, size_t rows,
                              void* stream) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (!out) return fail(PHE_HIP_EINVAL, "null buffer");
    if (int rc = bind_device(ctx)) return rc;
    PHE_CTX_ORDER(ctx, stream);
    return multiexp_rows_impl(ctx, base, base_inv, e, neg, exp_limbs, max_exp_bits, out, batch, rows, (hipStream_t)stream);
}

// out[r] = prod over the entries of row r of b[col]^exp (csrc/split_core.h: multiexp_tables_body + multiexp_lookup_body):
// the 2^w-ary tables of the whole vector once, then one ladder per row over that row's entries only.
int phe_hip_multiexp_csr_dev(phe_hip_ctx* ctx, const uint32_t* base, const uint32_t* base_inv, size_t batch,
                             const uint64_t* row_ptr, const uint32_t* cols, const uint32_t* e, const uint8_t* neg,
                             int exp_limbs, int max_exp_bits, const uint32_t* order, uint32_t* out, size_t rows,
                             void* stream) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (rows == 0) return PHE_HIP_OK;
    if (!out || !base || batch == 0) return fail(PHE_HIP_EINVAL, "null buffer / empty vector");
    if (!e || exp_limbs < 1) return fail(PHE_HIP_EINVAL, "null buffer / exp_limbs");
    if ((cols != nullptr) != (row_ptr != nullptr)) return fail(PHE_HIP_EINVAL, "row_ptr and cols come together (both NULL = dense rows)");
    if (neg && !base_inv) return fail(PHE_HIP_EINVAL, "a sign mask needs the inverted bases");
    if (max_exp_bits <= 0 || max_exp_bits > 32 * exp_limbs) max_exp_bits = 32 * exp_limbs;
    if (int rc = bind_device(ctx)) return rc;
    PHE_CTX_ORDER(ctx, stream);
    if (!(ctx->use_split && ctx->d_nsplit.G))
        return fail(PHE_HIP_EINVAL, "the table form needs the split-modulus engine (call phe_hip_multiexp_dev row by row on this key)");
    hipStream_t st = (hipStream_t)stream;
    const int w = host::pick_multi_window(max_exp_bits);
    const size_t per = ((size_t)1 << w) - 1, signs = base_inv ? 2 : 1;
    {   // tables: one limb group per (ciphertext, sign)
        // one geometry for both kernels (the table rows are H = G*L limbs of that geometry); the ladders are the bulk of
        // the work and their parallelism is `rows`: 16-lane groups (half the time per product) until the rows alone
        // fill the throughput geometry's resident groups halfway
        const DevSplit& M = pick_nsplit(ctx, rows);
        int rc = ensure_words(&ctx->lookup, &ctx->lookup_words, batch * signs * per * 2 * (size_t)M.H);
        if (rc) return rc;
        SplitTableArgs T;
        T.mod = M.c;
        T.base = base;
        T.base_inv = base_inv;
        T.base_limbs = ctx->pub.s2;
        T.base_chunks = chunks_for(ctx->pub.s2, M.rows);
        T.window = w;
        T.table = ctx->lookup;
        T.batch = batch;
        const int blocks = grid_blocks(ctx, batch * signs, M.G, 2);
        if (PHE_SPLIT_BY_GROUP(M.G, launch_multi_tables(M.L, blocks, st, T)) < 0)
            return fail(PHE_HIP_EINVAL, "unsupported split geometry");
        HIP_TRY(hipGetLastError());
        const DevSplit& M2 = M;
        SplitLookupArgs A;
        A.mod = M2.c;
        A.table = ctx->lookup;
        A.signs = (int)signs;
        A.row_ptr = row_ptr;
        A.cols = cols;
        A.exps = e;
        A.neg = neg;
        A.order = order;
        A.exp_limbs = exp_limbs;
        A.window = w;
        A.n_windows = std::max(1, (max_exp_bits + w - 1) / w);
        A.out = out;
        A.out_limbs = ctx->pub.s2;
        A.batch = batch;
        A.rows = rows;
        int per_cu = ctx->blocks_per_cu;
        if (per_cu == 0) per_cu = PHE_SPLIT_BY_GROUP(M2.G, occ_multi_split(M2.L));
        if (per_cu < 0) return fail(PHE_HIP_EINVAL, "unsupported split geometry");
        const int blocks2 = grid_blocks(ctx, rows, M2.G, per_cu);
        if (PHE_SPLIT_BY_GROUP(M2.G, launch_multi_lookup(M2.L, blocks2, st, A)) < 0)
            return fail(PHE_HIP_EINVAL, "unsupported split geometry");
        HIP_TRY(hipGetLastError());
    }
    return PHE_HIP_OK;
}

// ---- host-pointer entry points ------------------------------------------------------------------
// A call with a handful of rows (every operand and the result within kMappedSlotWords) skips the three blocking hipMemcpy of
// the path below — 36 of the 69 us of a one-row phe_hip_mulmod at 2048 bits (tools/scalar_op_breakdown.py,
// profiles/r03u_scalar_breakdown_2048.txt): operands are copied by the CPU into a pinned, device-mapped buffer, the kernels
// read them (and write the result) across PCIe, one stream synchronisation ends the call.  Scratch and tables stay in HBM.
static const size_t kMappedSlotWords = 8192;  // 32 KiB per slot: 64 rows of a 2048-bit ciphertext
static bool use_mapped(phe_hip_ctx* ctx, size_t w0, size_t w1, size_t w2) {
    if (ctx->no_mapped || std::max(w0, std::max(w1, w2)) > kMappedSlotWords) return false;
    if (!ctx->mapped_host) {
        void *h = nullptr, *d = nullptr;
        if (getenv("PHE_HIP_NO_MAPPED_STAGING") || hipHostMalloc(&h, 3 * kMappedSlotWords * 4, hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
            (void)hipGetLastError();
            if (h) (void)hipHostFree(h);
            ctx->no_mapped = true;
            return false;
        }
        ctx->mapped_host = (uint32_t*)h;
        ctx->mapped_dev = (uint32_t*)d;
    }
    return true;
}
// operand `slot` of a host-pointer call: ctx->cur[slot] is where the kernels find it (host_ptr == nullptr: room for a result)
static int stage_in(phe_hip_ctx* ctx, int slot, const uint32_t* host_ptr, size_t words, bool mapped = false) {
    if (mapped) {
        ctx->cur[slot] = ctx->mapped_dev + (size_t)slot * kMappedSlotWords;
        if (host_ptr) memcpy(ctx->mapped_host + (size_t)slot * kMappedSlotWords, host_ptr, words * 4);
        return PHE_HIP_OK;
    }
    int rc = ensure_words(&ctx->stage[slot], &ctx->stage_words[slot], words);
    if (rc) return rc;
    ctx->cur[slot] = ctx->stage[slot];
    if (host_ptr) HIP_TRY(hipMemcpy(ctx->stage[slot], host_ptr, words * 4, hipMemcpyHostToDevice));
    return PHE_HIP_OK;
}
// the result in `slot` back to the caller (rc: what the launches returned; a mapped call is drained either way, so that
// nothing still reads or writes the buffer when the next call fills it)
static int stage_out(phe_hip_ctx* ctx, int rc, int slot, void* host_out, size_t bytes, bool mapped) {
    if (mapped) {
        const hipError_t e = hipStreamSynchronize(nullptr);
        if (rc) return rc;
        if (e != hipSuccess) return fail(PHE_HIP_EHIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
        memcpy(host_out, ctx->mapped_host + (size_t)slot * kMappedSlotWords, bytes);
        return PHE_HIP_OK;
    }
    if (rc) return rc;
    HIP_TRY(hipMemcpy(host_out, ctx->stage[slot], bytes, hipMemcpyDeviceToHost));
    return PHE_HIP_OK;
}

// ---- the same entry points for LARGE host batches: a chunked pipeline inside the library ----------------------------
// The caller's buffers are ordinary (pageable) memory: a blocking hipMemcpy of the whole batch before and after the
// kernel leaves the GPU idle during both copies.  From two chunks on, a batch instead moves in chunks of one full
// residency of limb groups through PINNED staging buffers, two slots, three streams:
//     host copy in (CPU) -> H2D (s_in) -> kernel (s_comp) -> D2H (s_out) -> host copy out (CPU)
// so that the uploads of chunk k+1 and the downloads of chunk k-1 run under the kernel of chunk k.  A maintainer who
// binds only the plain host-pointer functions gets the overlap without writing a pipeline.
// Chunks are multiples of every kernel's resident limb groups (32768 / 65536): a short first chunk (little to upload
// before the first kernel starts), long middle chunks (each launch ends with a drain bubble: fewer launches), a short last
// one (little to download after the last kernel ends).
static const size_t kPipeEdgeRows = 65536, kPipeChunkRows = 131072;  // (decrypt keeps 65536 groups resident: an edge of 32768 rows would run at half occupancy)
static const size_t kPipeChunkBytes = (size_t)64 << 20;  // cap of one staging buffer: wide keys take proportionally fewer rows per chunk

static int pipe_setup(phe_hip_ctx* ctx) {
    auto& P = ctx->pipe;
    if (P.ready) return PHE_HIP_OK;
    HIP_TRY(hipStreamCreateWithFlags(&P.s_in, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&P.s_comp, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&P.s_out, hipStreamNonBlocking));
    for (int k = 0; k < 2; ++k) {
        HIP_TRY(hipEventCreateWithFlags(&P.ev_in[k], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&P.ev_comp[k], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&P.ev_out[k], hipEventDisableTiming));
    }
    P.ready = true;
    return PHE_HIP_OK;
}
static int pipe_buffers(phe_hip_ctx* ctx, int slot, int j, size_t words) {
    auto& P = ctx->pipe;
    if (P.pin_words[slot][j] < words) {
        if (P.pin[slot][j]) HIP_TRY(hipHostFree(P.pin[slot][j]));
        P.pin[slot][j] = nullptr;
        P.pin_words[slot][j] = 0;
        HIP_TRY(hipHostMalloc((void**)&P.pin[slot][j], words * 4, hipHostMallocDefault));
        P.pin_words[slot][j] = words;
    }
    return ensure_words(&P.dev[slot][j], &P.dev_words[slot][j], words);
}

// in0/in1: host rows of w0/w1 words (in1 may be null), out: host rows of wo words; launch(d_in0, d_in1, d_out, rows, stream)
extern "C++" {
template <class Launch>
static int run_pipelined(phe_hip_ctx* ctx, const uint32_t* in0, size_t w0, const uint32_t* in1, size_t w1, uint32_t* out,
                         size_t wo, size_t batch, Launch launch) {
    if (int rc = pipe_setup(ctx)) return rc;
    auto& P = ctx->pipe;
    // rows per chunk: the usual sizes, fewer when the rows are wide (a staging buffer stays below kPipeChunkBytes)
    const size_t row_bytes = 4 * std::max(std::max(w0, w1), wo);
    size_t chunk_rows = kPipeChunkRows, edge_rows = kPipeEdgeRows;
    while (chunk_rows > 8192 && chunk_rows * row_bytes > kPipeChunkBytes) {
        chunk_rows /= 2;
        edge_rows = std::max<size_t>(edge_rows / 2, 4096);
    }
    // chunk boundaries: edge | full chunks ... | edge
    std::vector<size_t> lo;
    size_t widest = 0;
    {
        size_t at = 0;
        lo.push_back(0);
        at = std::min(batch, edge_rows);
        while (at < batch) {
            lo.push_back(at);
            const size_t left = batch - at;
            at += (left > chunk_rows + edge_rows) ? chunk_rows : (left > edge_rows ? left - edge_rows : left);
        }
        lo.push_back(batch);
        for (size_t k = 0; k + 1 < lo.size(); ++k) widest = std::max(widest, lo[k + 1] - lo[k]);
    }
    const size_t n_chunks = lo.size() - 1;
    for (int slot = 0; slot < 2; ++slot) {
        int rc = pipe_buffers(ctx, slot, 0, widest * w0);
        if (!rc && in1) rc = pipe_buffers(ctx, slot, 1, widest * w1);
        if (!rc) rc = pipe_buffers(ctx, slot, 2, widest * wo);
        if (rc) return rc;
    }
    auto rows_of = [&](size_t k) { return lo[k + 1] - lo[k]; };
    auto drain = [&](size_t k) -> int {  // chunk k's results: wait for its download, then hand them to the caller
        const int slot = (int)(k & 1);
        HIP_TRY(hipEventSynchronize(P.ev_out[slot]));
        memcpy(out + lo[k] * wo, P.pin[slot][2], rows_of(k) * wo * 4);
        return PHE_HIP_OK;
    };
    // one chunk through the three streams; a failure leaves copies / kernels of earlier chunks in flight on the pinned and
    // device slots, so the streams are drained below before the error is reported
    auto step = [&](size_t k) -> int {
        const int slot = (int)(k & 1);
        const size_t rows = rows_of(k);
        if (k >= 2)
            if (int rc = drain(k - 2)) return rc;  // frees this slot's buffers (its kernel and copies are complete)
        memcpy(P.pin[slot][0], in0 + lo[k] * w0, rows * w0 * 4);
        HIP_TRY(hipMemcpyAsync(P.dev[slot][0], P.pin[slot][0], rows * w0 * 4, hipMemcpyHostToDevice, P.s_in));
        if (in1) {
            memcpy(P.pin[slot][1], in1 + lo[k] * w1, rows * w1 * 4);
            HIP_TRY(hipMemcpyAsync(P.dev[slot][1], P.pin[slot][1], rows * w1 * 4, hipMemcpyHostToDevice, P.s_in));
        }
        HIP_TRY(hipEventRecord(P.ev_in[slot], P.s_in));
        HIP_TRY(hipStreamWaitEvent(P.s_comp, P.ev_in[slot], 0));
        if (int rc = launch(P.dev[slot][0], in1 ? P.dev[slot][1] : nullptr, P.dev[slot][2], rows, P.s_comp)) return rc;
        HIP_TRY(hipEventRecord(P.ev_comp[slot], P.s_comp));
        HIP_TRY(hipStreamWaitEvent(P.s_out, P.ev_comp[slot], 0));
        HIP_TRY(hipMemcpyAsync(P.pin[slot][2], P.dev[slot][2], rows * wo * 4, hipMemcpyDeviceToHost, P.s_out));
        HIP_TRY(hipEventRecord(P.ev_out[slot], P.s_out));
        return PHE_HIP_OK;
    };
    int rc = PHE_HIP_OK;
    for (size_t k = 0; k < n_chunks && !rc; ++k) rc = step(k);
    for (size_t k = (n_chunks >= 2 ? n_chunks - 2 : 0); k < n_chunks && !rc; ++k) rc = drain(k);
    if (rc) {
        const std::string msg = g_err;  // the first error is the one to report
        (void)hipStreamSynchronize(P.s_in);
        (void)hipStreamSynchronize(P.s_comp);
        (void)hipStreamSynchronize(P.s_out);
        g_err = msg;
        return rc;
    }
    ctx->last_path |= kPathPipelined;
    return PHE_HIP_OK;
}
}  // extern "C++"
static bool pipelined_batch(size_t batch) { return batch >= 2 * kPipeEdgeRows && !getenv("PHE_HIP_NO_PIPELINE"); }

// Give the grow-only buffers of a context back (window tables, intermediates, the staging of the host-pointer entry points
// and the pinned chunk buffers of their pipeline): after a one-off large batch a long-lived context need not keep GiBs.
// Synchronises the device; the next call allocates what it needs again.
int phe_hip_ctx_release_scratch(phe_hip_ctx* ctx) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (int rc = bind_device(ctx)) return rc;
    HIP_TRY(hipDeviceSynchronize());
    uint32_t** bufs[] = {&ctx->table, &ctx->table2, &ctx->scratch, &ctx->partial, &ctx->lookup, &ctx->unit_tmp,
                         &ctx->stage[0], &ctx->stage[1], &ctx->stage[2], &ctx->item_sched};
    size_t* sizes[] = {&ctx->table_words, &ctx->table2_words, &ctx->scratch_words, &ctx->partial_words, &ctx->lookup_words,
                       &ctx->unit_tmp_words, &ctx->stage_words[0], &ctx->stage_words[1], &ctx->stage_words[2], &ctx->item_sched_words};
    for (int i = 0; i < 10; ++i) {
        if (*bufs[i]) HIP_TRY(hipFree(*bufs[i]));
        *bufs[i] = nullptr;
        *sizes[i] = 0;
    }
    for (int k = 0; k < 2; ++k)
        for (int j = 0; j < 3; ++j) {
            if (ctx->pipe.pin[k][j]) HIP_TRY(hipHostFree(ctx->pipe.pin[k][j]));
            if (ctx->pipe.dev[k][j]) HIP_TRY(hipFree(ctx->pipe.dev[k][j]));
            ctx->pipe.pin[k][j] = ctx->pipe.dev[k][j] = nullptr;
            ctx->pipe.pin_words[k][j] = ctx->pipe.dev_words[k][j] = 0;
        }
    return PHE_HIP_OK;
}


int phe_hip_encrypt(phe_hip_ctx* ctx, const uint32_t* m, const uint32_t* r, uint32_t* c, size_t batch) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (batch == 0) return PHE_HIP_OK;
    if (!m || !r || !c) return fail(PHE_HIP_EINVAL, "null buffer");
    if (int rc = bind_device(ctx)) return rc;
    const size_t s1 = (size_t)ctx->pub.s1, s2 = (size_t)ctx->pub.s2;
    if (pipelined_batch(batch))
        return run_pipelined(ctx, m, s1, r, s1, c, s2, batch,
                             [&](uint32_t* d0, uint32_t* d1, uint32_t* d2, size_t rows, hipStream_t st) {
                                 return phe_hip_encrypt_dev(ctx, d0, d1, d2, rows, st);
                             });
    const bool mp = use_mapped(ctx, batch * s1, batch * s1, batch * s2);
    int rc = stage_in(ctx, 0, m, batch * s1, mp);
    if (!rc) rc = stage_in(ctx, 1, r, batch * s1, mp);
    if (!rc) rc = stage_in(ctx, 2, nullptr, batch * s2, mp);
    if (!rc) rc = phe_hip_encrypt_dev(ctx, ctx->cur[0], ctx->cur[1], ctx->cur[2], batch, nullptr);
    return stage_out(ctx, rc, 2, c, batch * s2 * 4, mp);
}

int phe_hip_encrypt_owner(phe_hip_ctx* ctx, const uint32_t* m, const uint32_t* r, uint32_t* c, size_t batch) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (batch == 0) return PHE_HIP_OK;
    if (!m || !r || !c) return fail(PHE_HIP_EINVAL, "null buffer");
    if (int rc = bind_device(ctx)) return rc;
    const size_t s1 = (size_t)ctx->pub.s1, s2 = (size_t)ctx->pub.s2;
    if (pipelined_batch(batch))
        return run_pipelined(ctx, m, s1, r, s1, c, s2, batch,
                             [&](uint32_t* d0, uint32_t* d1, uint32_t* d2, size_t rows, hipStream_t st) {
                                 return phe_hip_encrypt_owner_dev(ctx, d0, d1, d2, rows, st);
                             });
    const bool mp = use_mapped(ctx, batch * s1, batch * s1, batch * s2);
    int rc = stage_in(ctx, 0, m, batch * s1, mp);
    if (!rc) rc = stage_in(ctx, 1, r, batch * s1, mp);
    if (!rc) rc = stage_in(ctx, 2, nullptr, batch * s2, mp);
    if (!rc) rc = phe_hip_encrypt_owner_dev(ctx, ctx->cur[0], ctx->cur[1], ctx->cur[2], batch, nullptr);
    return stage_out(ctx, rc, 2, c, batch * s2 * 4, mp);
}

int phe_hip_obfuscate(phe_hip_ctx* ctx, const uint32_t* c_in, const uint32_t* r, uint32_t* c_out, size_t batch) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (batch == 0) return PHE_HIP_OK;
    if (!c_in || !r || !c_out) return fail(PHE_HIP_EINVAL, "null buffer");
    if (int rc = bind_device(ctx)) return rc;
    const size_t s1 = (size_t)ctx->pub.s1, s2 = (size_t)ctx->pub.s2;
    if (pipelined_batch(batch))
        return run_pipelined(ctx, c_in, s2, r, s1, c_out, s2, batch,
                             [&](uint32_t* d0, uint32_t* d1, uint32_t* d2, size_t rows, hipStream_t st) {
                                 return phe_hip_obfuscate_dev(ctx, d0, d1, d2, rows, st);
                             });
    const bool mp = use_mapped(ctx, batch * s2, batch * s1, batch * s2);
    int rc = stage_in(ctx, 0, c_in, batch * s2, mp);
    if (!rc) rc = stage_in(ctx, 1, r, batch * s1, mp);
    if (!rc) rc = stage_in(ctx, 2, nullptr, batch * s2, mp);
    if (!rc) rc = phe_hip_obfuscate_dev(ctx, ctx->cur[0], ctx->cur[1], ctx->cur[2], batch, nullptr);
    return stage_out(ctx, rc, 2, c_out, batch * s2 * 4, mp);
}

int phe_hip_decrypt(phe_hip_ctx* ctx, const uint32_t* c, uint32_t* m, size_t batch) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (batch == 0) return PHE_HIP_OK;
    if (!c || !m) return fail(PHE_HIP_EINVAL, "null buffer");
    if (int rc = bind_device(ctx)) return rc;
    const size_t s1 = (size_t)ctx->pub.s1, s2 = (size_t)ctx->pub.s2;
    if (pipelined_batch(batch))
        return run_pipelined(ctx, c, s2, nullptr, 0, m, s1, batch,
                             [&](uint32_t* d0, uint32_t*, uint32_t* d2, size_t rows, hipStream_t st) {
                                 return phe_hip_decrypt_dev(ctx, d0, d2, rows, st);
                             });
    const bool mp = use_mapped(ctx, batch * s2, batch * s1, 0);
    int rc = stage_in(ctx, 0, c, batch * s2, mp);
    if (!rc) rc = stage_in(ctx, 1, nullptr, batch * s1, mp);
    if (!rc) rc = phe_hip_decrypt_dev(ctx, ctx->cur[0], ctx->cur[1], batch, nullptr);
    return stage_out(ctx, rc, 1, m, batch * s1 * 4, mp);
}

int phe_hip_mulmod(phe_hip_ctx* ctx, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t batch) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (batch == 0) return PHE_HIP_OK;
    if (!a || !b || !out) return fail(PHE_HIP_EINVAL, "null buffer");
    if (int rc = bind_device(ctx)) return rc;
    const size_t s2 = (size_t)ctx->pub.s2;
    // (no chunked pipeline here: one product per 1.5 KB moved is PCIe-bound, and the runtime's own pageable-memory copy
    //  (24 GB/s) beats a single-threaded copy into pinned staging (19 GB/s measured, profiles/r02c_host_abi.json))
    const bool mp = use_mapped(ctx, batch * s2, batch * s2, batch * s2);
    int rc = stage_in(ctx, 0, a, batch * s2, mp);
    if (!rc) rc = stage_in(ctx, 1, b, batch * s2, mp);
    if (!rc) rc = stage_in(ctx, 2, nullptr, batch * s2, mp);
    if (!rc) rc = phe_hip_mulmod_dev(ctx, ctx->cur[0], ctx->cur[1], ctx->cur[2], batch, nullptr);
    return stage_out(ctx, rc, 2, out, batch * s2 * 4, mp);
}

int phe_hip_add_plain(phe_hip_ctx* ctx, const uint32_t* c, const uint32_t* m, uint32_t* out, size_t batch) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (batch == 0) return PHE_HIP_OK;
    if (!c || !m || !out) return fail(PHE_HIP_EINVAL, "null buffer");
    if (int rc = bind_device(ctx)) return rc;
    const size_t s1 = (size_t)ctx->pub.s1, s2 = (size_t)ctx->pub.s2;
    const bool mp = use_mapped(ctx, batch * s2, batch * s1, batch * s2);
    int rc = stage_in(ctx, 0, c, batch * s2, mp);
    if (!rc) rc = stage_in(ctx, 1, m, batch * s1, mp);
    if (!rc) rc = stage_in(ctx, 2, nullptr, batch * s2, mp);
    if (!rc) rc = phe_hip_add_plain_dev(ctx, ctx->cur[0], ctx->cur[1], ctx->cur[2], batch, nullptr);
    return stage_out(ctx, rc, 2, out, batch * s2 * 4, mp);
}

int phe_hip_powmod(phe_hip_ctx* ctx, const uint32_t* base, const uint32_t* e, int exp_limbs, uint32_t* out, size_t batch) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (batch == 0) return PHE_HIP_OK;
    if (!base || !e || !out || exp_limbs < 1) return fail(PHE_HIP_EINVAL, "null buffer / exp_limbs");
    if (int rc = bind_device(ctx)) return rc;
    int max_bits = 0;
    for (size_t i = 0; i < batch; ++i) {
        const uint32_t* row = e + i * (size_t)exp_limbs;
        for (int k = exp_limbs - 1; k >= 0; --k)
            if (row[k]) {
                max_bits = std::max(max_bits, 32 * k + 32 - __builtin_clz(row[k]));
                break;
            }
    }
    if (max_bits == 0) max_bits = 1;
    const size_t s2 = (size_t)ctx->pub.s2;
    const bool mp = use_mapped(ctx, batch * s2, batch * (size_t)exp_limbs, batch * s2);
    int rc = stage_in(ctx, 0, base, batch * s2, mp);
    if (!rc) rc = stage_in(ctx, 2, nullptr, batch * s2, mp);
    if (rc) return rc;
    // A handful of numbers (EncryptedNumber.__mul__, one at a time): the exponents are on the host here, so every number gets
    // its OWN sliding-window schedule and runs on a pair of wavefronts like a scalar encrypt / decrypt does (the per-element
    // kernel takes fixed windows over the longest exponent on one wave per number).  A zero exponent keeps the general path.
    if (const DevSplit& sp = pick_nsplit(ctx, batch); ctx->use_split && sp.G == 64 && ab_offered(ctx, sp, batch, 1)) {
        std::vector<uint32_t> blob, meta;
        DevSchedule E;
        E.n_ops = 0;
        E.tbl_entries = 1;
        bool ok = true;
        for (size_t i = 0; i < batch && ok; ++i) {
            const Big ei = host::big_from(e + i * (size_t)exp_limbs, exp_limbs, exp_limbs);
            if (host::big_bits(ei) == 0) {
                ok = false;
                break;
            }
            const host::Schedule S = host::build_schedule(ei);
            meta.push_back((uint32_t)S.ops.size());
            meta.push_back((uint32_t)S.first_idx);
            meta.push_back((uint32_t)S.tbl_entries);
            meta.push_back((uint32_t)blob.size());
            blob.insert(blob.end(), S.ops.begin(), S.ops.end());
            E.n_ops = std::max(E.n_ops, (int)S.ops.size());
            E.tbl_entries = std::max(E.tbl_entries, S.tbl_entries);
        }
        if (ok) {
            const size_t ops_words = std::max<size_t>(1, blob.size());
            blob.resize(ops_words);
            blob.insert(blob.end(), meta.begin(), meta.end());
            rc = ensure_words(&ctx->item_sched, &ctx->item_sched_words, blob.size());
            if (rc) return rc;
            HIP_TRY(hipMemcpy(ctx->item_sched, blob.data(), blob.size() * 4, hipMemcpyHostToDevice));
            E.ops = ctx->item_sched;
            E.first_idx = 0;
            PHE_CTX_ORDER(ctx, nullptr);
            ctx->last_path = kPathWavePairs;
            ctx->last_geom_pub = geom_code(sp.G, sp.L);
            rc = launch_split_ab(ctx, kModeEncrypt, sp, E, nullptr, nullptr, ctx->cur[0], ctx->pub.s2, nullptr, 0, ctx->cur[2], nullptr,
                                 ctx->pub.s2, batch, nullptr, ctx->item_sched + ops_words);
            return stage_out(ctx, rc, 2, out, batch * s2 * 4, mp);
        }
    }
    rc = stage_in(ctx, 1, e, batch * (size_t)exp_limbs, mp);
    if (!rc) rc = phe_hip_powmod_dev(ctx, ctx->cur[0], ctx->cur[1], exp_limbs, max_bits, ctx->cur[2], batch, nullptr);
    return stage_out(ctx, rc, 2, out, batch * s2 * 4, mp);
}

static int max_exp_bits_of(const uint32_t* e, int exp_limbs, size_t batch) {
    int max_bits = 0;
    for (size_t i = 0; i < batch; ++i) {
        const uint32_t* row = e + i * (size_t)exp_limbs;
        for (int k = exp_limbs - 1; k >= 0; --k)
            if (row[k]) {
                max_bits = std::max(max_bits, 32 * k + 32 - __builtin_clz(row[k]));
                break;
            }
    }
    return max_bits == 0 ? 1 : max_bits;
}

int phe_hip_multiexp(phe_hip_ctx* ctx, const uint32_t* base, const uint32_t* e, int exp_limbs, uint32_t* out, size_t batch) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (!out) return fail(PHE_HIP_EINVAL, "null buffer");
    if (batch && (!base || !e || exp_limbs < 1)) return fail(PHE_HIP_EINVAL, "null buffer / exp_limbs");
    if (int rc = bind_device(ctx)) return rc;
    const size_t s2 = (size_t)ctx->pub.s2;
    int rc = stage_in(ctx, 0, batch ? base : nullptr, std::max<size_t>(1, batch) * s2);
    if (!rc) rc = stage_in(ctx, 1, batch ? e : nullptr, std::max<size_t>(1, batch * (size_t)std::max(1, exp_limbs)));
    if (!rc) rc = stage_in(ctx, 2, nullptr, s2);
    if (!rc)
        rc = phe_hip_multiexp_dev(ctx, ctx->stage[0], ctx->stage[1], exp_limbs, batch ? max_exp_bits_of(e, exp_limbs, batch) : 1,
                                  ctx->stage[2], batch, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out, ctx->stage[2], s2 * 4, hipMemcpyDeviceToHost));
    return PHE_HIP_OK;
}

// ---- batched inversion (Montgomery's trick over a product tree) ------------------------------------
static bool host_invert(const Big& a_in, const Big& N, Big& out) { return host::big_invert_odd(a_in, N, out); }

// a_is_device/out_is_device pick the copy kinds; the trees live in stage[0] (products) and stage[1] (inverses).
static int invert_impl(phe_hip_ctx* ctx, const uint32_t* a, bool a_is_device, uint32_t* out, bool out_is_device,
                       size_t batch, size_t* bad_index, hipStream_t st) {
    const int s2 = ctx->pub.s2;
    const size_t w = (size_t)s2;
    const DevModulus& M = ctx->d_nsq;
    // level sizes of the product tree
    std::vector<size_t> cnt{batch};
    while (cnt.back() > 1) cnt.push_back((cnt.back() + 1) / 2);
    std::vector<size_t> off(cnt.size());
    size_t total = 0;
    for (size_t k = 0; k < cnt.size(); ++k) { off[k] = total; total += cnt[k]; }
    int rc = stage_in(ctx, 0, nullptr, total * w);
    if (!rc) rc = stage_in(ctx, 1, nullptr, total * w);
    if (rc) return rc;
    uint32_t* prod = ctx->stage[0];
    uint32_t* inv = ctx->stage[1];
    HIP_TRY(hipMemcpyAsync(prod, a, batch * w * 4, a_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    for (size_t k = 0; k + 1 < cnt.size(); ++k) {
        const size_t pairs = cnt[k] / 2;
        uint32_t* src = prod + off[k] * w;
        uint32_t* dst = prod + off[k + 1] * w;
        if (pairs) {
            rc = launch_mul(ctx, M, src, 2 * w, src + w, 2 * w, dst, w, s2, pairs, st);
            if (rc) return rc;
        }
        if (cnt[k] & 1) HIP_TRY(hipMemcpyAsync(dst + pairs * w, src + (cnt[k] - 1) * w, w * 4, hipMemcpyDeviceToDevice, st));
    }
    // one scalar inversion of the root on the host
    Big root(w), N = ctx->pub.nsq32, rinv;
    HIP_TRY(hipMemcpyAsync(root.data(), prod + off.back() * w, w * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const bool reduced_ok = host::big_cmp(root, N) < 0;
    if (!reduced_ok || !host_invert(root, N, rinv)) {
        // slow path only on failure: find the first row that is not a unit (or not reduced)
        std::vector<uint32_t> h(batch * w);
        HIP_TRY(hipMemcpy(h.data(), prod, batch * w * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < batch; ++i) {
            Big x(h.begin() + (long)(i * w), h.begin() + (long)((i + 1) * w)), tmp;
            Big xr = host::big_cmp(x, N) < 0 ? x : host::big_mod(x, N);
            if (!host_invert(xr, N, tmp)) {
                if (bad_index) *bad_index = i;
                return fail(PHE_HIP_ENOINVERSE, "invert() no inverse exists");
            }
        }
        return fail(PHE_HIP_EINVAL, "inversion failed although every element is a unit (operands must be < n^2)");
    }
    HIP_TRY(hipMemcpyAsync(inv + off.back() * w, rinv.data(), w * 4, hipMemcpyHostToDevice, st));
    for (size_t k = cnt.size() - 1; k-- > 0;) {
        const size_t pairs = cnt[k] / 2;
        uint32_t* p = prod + off[k] * w;
        uint32_t* up = inv + off[k + 1] * w;
        uint32_t* dn = inv + off[k] * w;
        if (pairs) {
            // inv[2i] = up[i] * prod[2i+1];  inv[2i+1] = up[i] * prod[2i]
            rc = launch_mul(ctx, M, up, w, p + w, 2 * w, dn, 2 * w, s2, pairs, st);
            if (!rc) rc = launch_mul(ctx, M, up, w, p, 2 * w, dn + w, 2 * w, s2, pairs, st);
            if (rc) return rc;
        }
        if (cnt[k] & 1) HIP_TRY(hipMemcpyAsync(dn + (cnt[k] - 1) * w, up + pairs * w, w * 4, hipMemcpyDeviceToDevice, st));
    }
    HIP_TRY(hipMemcpyAsync(out, inv, batch * w * 4, out_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));  // rinv (host) was an async source; results are complete on return
    return PHE_HIP_OK;
}

int phe_hip_invert(phe_hip_ctx* ctx, const uint32_t* a, uint32_t* out, size_t batch, size_t* bad_index) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (batch == 0) return PHE_HIP_OK;
    if (!a || !out) return fail(PHE_HIP_EINVAL, "null buffer");
    if (int rc = bind_device(ctx)) return rc;
    return invert_impl(ctx, a, false, out, false, batch, bad_index, nullptr);
}

int phe_hip_invert_dev(phe_hip_ctx* ctx, const uint32_t* a, uint32_t* out, size_t batch, size_t* bad_index, void* stream) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (batch == 0) return PHE_HIP_OK;
    if (!a || !out) return fail(PHE_HIP_EINVAL, "null buffer");
    if (int rc = bind_device(ctx)) return rc;
    PHE_CTX_ORDER(ctx, stream);
    return invert_impl(ctx, a, true, out, true, batch, bad_index, (hipStream_t)stream);
}

int phe_hip_select_rows_dev(phe_hip_ctx* ctx, const uint32_t* a, const uint32_t* b, const uint8_t* mask, uint32_t* out,
                            int limbs, size_t batch, void* stream) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (batch == 0) return PHE_HIP_OK;
    if (!a || !b || !mask || !out || limbs < 1) return fail(PHE_HIP_EINVAL, "null buffer / limbs");
    if (int rc = bind_device(ctx)) return rc;
    PHE_CTX_ORDER(ctx, stream);  // (ordered against the context's earlier work on other streams, like every *_dev entry point)
    const size_t total = batch * (size_t)limbs;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, (size_t)ctx->n_cus * 8);
    k_select_rows<<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(a, b, mask, out, limbs, (uint64_t)batch);
    HIP_TRY(hipGetLastError());
    return PHE_HIP_OK;
}

static int move_rows(phe_hip_ctx* ctx, bool scatter, const uint32_t* src, const uint32_t* idx, uint32_t* dst, int limbs, size_t count,
                     size_t far_rows, void* stream) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (count == 0) return PHE_HIP_OK;
    if (!src || !idx || !dst || limbs < 1) return fail(PHE_HIP_EINVAL, "null buffer / limbs");
    if (int rc = bind_device(ctx)) return rc;
    PHE_CTX_ORDER(ctx, stream);
    const size_t total = count * (size_t)limbs;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, (size_t)ctx->n_cus * 8);
    if (scatter) k_move_rows<true><<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(src, idx, dst, limbs, (uint64_t)count, (uint64_t)far_rows);
    else k_move_rows<false><<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(src, idx, dst, limbs, (uint64_t)count, (uint64_t)far_rows);
    HIP_TRY(hipGetLastError());
    return PHE_HIP_OK;
}
int phe_hip_gather_rows_dev(phe_hip_ctx* ctx, const uint32_t* src, size_t src_rows, const uint32_t* idx, uint32_t* dst, int limbs,
                            size_t count, void* stream) {
    return move_rows(ctx, false, src, idx, dst, limbs, count, src_rows, stream);
}
int phe_hip_scatter_rows_dev(phe_hip_ctx* ctx, const uint32_t* src, const uint32_t* idx, uint32_t* dst, size_t dst_rows, int limbs,
                             size_t count, void* stream) {
    return move_rows(ctx, true, src, idx, dst, limbs, count, dst_rows, stream);
}

// ---- decimal wire format (csrc/radix_conv.h, kernels_radix.hip) -------------------------------------------------
static const size_t kMaxRadixTile = 160 * 1024;  // one LDS tile of 64 numbers must fit a CU

static int radix_flags(phe_hip_ctx* ctx, hipStream_t st) {
    if (!ctx->flags) HIP_TRY(hipMalloc((void**)&ctx->flags, 2 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(ctx->flags, 0xff, 2 * sizeof(unsigned long long), st));
    return PHE_HIP_OK;
}

int phe_hip_decimal_width(int words) { return words < 1 ? 0 : phe::decimal_width(words); }

int phe_hip_to_decimal_dev(phe_hip_ctx* ctx, const uint32_t* limbs, int words, char* digits, int width, size_t batch,
                           void* stream) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (batch == 0) return PHE_HIP_OK;
    if (!limbs || !digits || words < 1 || width < 1) return fail(PHE_HIP_EINVAL, "null buffer / words / width");
    if (phe::radix::tile_bytes(words) > kMaxRadixTile) return fail(PHE_HIP_EINVAL, "number too wide for the conversion tile");
    if (int rc = bind_device(ctx)) return rc;
    PHE_CTX_ORDER(ctx, stream);
    hipStream_t st = (hipStream_t)stream;
    if (int rc = radix_flags(ctx, st)) return rc;
    if (phe::radix::launch_to_decimal(limbs, words, digits, width, batch, ctx->flags, ctx->n_cus * 8, st) < 0)
        return fail(PHE_HIP_EHIP, "cannot size the conversion tile");
    HIP_TRY(hipGetLastError());
    unsigned long long bad[2];
    HIP_TRY(hipMemcpyAsync(bad, ctx->flags, sizeof bad, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (bad[0] != ~0ull) return fail(PHE_HIP_EINVAL, "row " + std::to_string(bad[0]) + " needs more than `width` digits");
    return PHE_HIP_OK;
}

int phe_hip_from_decimal_dev(phe_hip_ctx* ctx, const char* digits, int width, uint32_t* limbs, int words, size_t batch,
                             size_t* bad_index, void* stream) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (batch == 0) return PHE_HIP_OK;
    if (!limbs || !digits || words < 1 || width < 1) return fail(PHE_HIP_EINVAL, "null buffer / words / width");
    if (phe::radix::tile_bytes(words) > kMaxRadixTile) return fail(PHE_HIP_EINVAL, "number too wide for the conversion tile");
    if (int rc = bind_device(ctx)) return rc;
    PHE_CTX_ORDER(ctx, stream);
    hipStream_t st = (hipStream_t)stream;
    if (int rc = radix_flags(ctx, st)) return rc;
    if (phe::radix::launch_from_decimal(digits, width, limbs, words, batch, ctx->flags, ctx->flags + 1, ctx->n_cus * 8, st) < 0)
        return fail(PHE_HIP_EHIP, "cannot size the conversion tile");
    HIP_TRY(hipGetLastError());
    unsigned long long bad[2];
    HIP_TRY(hipMemcpyAsync(bad, ctx->flags, sizeof bad, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (bad[0] != ~0ull || bad[1] != ~0ull) {
        const bool is_char = bad[0] <= bad[1];
        if (bad_index) *bad_index = (size_t)(is_char ? bad[0] : bad[1]);
        return fail(PHE_HIP_EINVAL, is_char ? "invalid literal: not a decimal digit" : "value does not fit the limb width");
    }
    return PHE_HIP_OK;
}

int phe_hip_to_decimal(phe_hip_ctx* ctx, const uint32_t* limbs, int words, char* digits, int width, size_t batch) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (batch == 0) return PHE_HIP_OK;
    if (!limbs || !digits || words < 1 || width < 1) return fail(PHE_HIP_EINVAL, "null buffer / words / width");
    if (int rc = bind_device(ctx)) return rc;
    const size_t dwords = (batch * (size_t)width + 3) / 4;
    int rc = stage_in(ctx, 0, limbs, batch * (size_t)words);
    if (!rc) rc = stage_in(ctx, 1, nullptr, dwords);
    if (!rc) rc = phe_hip_to_decimal_dev(ctx, ctx->stage[0], words, (char*)ctx->stage[1], width, batch, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(digits, ctx->stage[1], batch * (size_t)width, hipMemcpyDeviceToHost));
    return PHE_HIP_OK;
}

int phe_hip_from_decimal(phe_hip_ctx* ctx, const char* digits, int width, uint32_t* limbs, int words, size_t batch,
                         size_t* bad_index) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (batch == 0) return PHE_HIP_OK;
    if (!limbs || !digits || words < 1 || width < 1) return fail(PHE_HIP_EINVAL, "null buffer / words / width");
    if (int rc = bind_device(ctx)) return rc;
    const size_t dwords = (batch * (size_t)width + 3) / 4;
    int rc = stage_in(ctx, 1, nullptr, dwords);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(ctx->stage[1], digits, batch * (size_t)width, hipMemcpyHostToDevice));
    rc = stage_in(ctx, 0, nullptr, batch * (size_t)words);
    if (!rc) rc = phe_hip_from_decimal_dev(ctx, (const char*)ctx->stage[1], width, ctx->stage[0], words, batch, bad_index, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(limbs, ctx->stage[0], batch * (size_t)words * 4, hipMemcpyDeviceToHost));
    return PHE_HIP_OK;
}

// ---- batched Miller-Rabin, one modulus per row (csrc/primality.h, kernels_mr.hip) ---------------------------------
int phe_hip_miller_rabin(int device, const uint32_t* n, const uint32_t* base, int limbs, uint8_t* pass, size_t batch) {
    if (batch == 0) return PHE_HIP_OK;
    if (!n || !base || !pass || limbs < 1) return fail(PHE_HIP_EINVAL, "null buffer / limbs");
    const host::Geometry geo = host::pick_geometry(32 * limbs, 0, 16);
    if (geo.G != 16) return fail(PHE_HIP_EINVAL, "candidates too wide for the compiled 16-lane kernels");
    for (size_t i = 0; i < batch; ++i) {  // n odd and > 3, 2 <= base <= n - 2 (what the callers' small-number paths leave)
        const host::Big N = host::big_from(n + i * (size_t)limbs, limbs, limbs);
        host::Big a = host::big_from(base + i * (size_t)limbs, limbs, limbs);
        host::Big two((size_t)limbs, 0u);
        two[0] = 2;
        if ((N[0] & 1u) == 0u || host::big_bits(N) < 3)
            return fail(PHE_HIP_EINVAL, "row " + std::to_string(i) + ": candidate must be odd and > 3");
        if (host::big_cmp(a, two) < 0) return fail(PHE_HIP_EINVAL, "row " + std::to_string(i) + ": base must be >= 2");
        host::big_add_inplace(a, two);  // base + 2 <= n, no overflow of the width because base < 2^(32 limbs) - 2 is implied below
        if (host::big_cmp(a, two) < 0 || host::big_cmp(a, N) > 0)
            return fail(PHE_HIP_EINVAL, "row " + std::to_string(i) + ": base must be <= n - 2");
    }
    HIP_TRY(hipSetDevice(device));
    uint32_t *d_n = nullptr, *d_b = nullptr;
    uint8_t* d_p = nullptr;
    const size_t words = batch * (size_t)limbs;
    auto cleanup = [&]() {
        if (d_n) (void)hipFree(d_n);
        if (d_b) (void)hipFree(d_b);
        if (d_p) (void)hipFree(d_p);
    };
    hipError_t e = hipMalloc((void**)&d_n, words * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&d_b, words * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&d_p, batch);
    if (e == hipSuccess) e = hipMemcpy(d_n, n, words * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_b, base, words * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        MillerRabinArgs A;
        A.n = d_n;
        A.base = d_b;
        A.limbs = limbs;
        A.pass = d_p;
        A.batch = batch;
        const int blocks = (int)std::min<size_t>((batch + 15) / 16, 65535);
        if (phe::mr::launch(geo.L, blocks, nullptr, A) < 0) {
            cleanup();
            return fail(PHE_HIP_EINVAL, "unsupported limb-group geometry");
        }
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(pass, d_p, batch, hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess) return fail(PHE_HIP_EHIP, std::string("miller_rabin: ") + hipGetErrorString(e));
    return PHE_HIP_OK;
}

// ---- memory helpers ---------------------------------------------------------------------------------
int phe_hip_malloc(phe_hip_ctx* ctx, size_t bytes, void** dptr) {
    if (check_ctx(ctx) || !dptr) return fail(PHE_HIP_EINVAL, "null argument");
    if (int rc = bind_device(ctx)) return rc;
    HIP_TRY(hipMalloc(dptr, bytes ? bytes : 4));
    return PHE_HIP_OK;
}
int phe_hip_free(phe_hip_ctx* ctx, void* dptr) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (int rc = bind_device(ctx)) return rc;
    if (dptr) HIP_TRY(hipFree(dptr));
    return PHE_HIP_OK;
}
int phe_hip_memcpy_h2d(phe_hip_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (int rc = bind_device(ctx)) return rc;
    HIP_TRY(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
    return PHE_HIP_OK;
}
int phe_hip_memcpy_d2h(phe_hip_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (int rc = bind_device(ctx)) return rc;
    HIP_TRY(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
    return PHE_HIP_OK;
}
int phe_hip_memcpy_d2d(phe_hip_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes, void* stream) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (int rc = bind_device(ctx)) return rc;
    HIP_TRY(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return PHE_HIP_OK;
}
// A stream that does not synchronise with the NULL stream: kernels queued on it keep running while the host uploads
// the next operands with the blocking copies above (which are ordered on the NULL stream).
int phe_hip_stream_create(phe_hip_ctx* ctx, void** stream) {
    if (check_ctx(ctx) || !stream) return fail(PHE_HIP_EINVAL, "null argument");
    if (int rc = bind_device(ctx)) return rc;
    hipStream_t st = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *stream = (void*)st;
    return PHE_HIP_OK;
}
int phe_hip_stream_destroy(phe_hip_ctx* ctx, void* stream) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (!stream) return PHE_HIP_OK;
    if (int rc = bind_device(ctx)) return rc;
    if (ctx->busy_valid && ctx->busy_stream == (hipStream_t)stream) {  // the context's last work ran here: finish it, forget the stream
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        ctx->busy_valid = false;
    }
    HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return PHE_HIP_OK;
}

int phe_hip_stream_sync(phe_hip_ctx* ctx, void* stream) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (int rc = bind_device(ctx)) return rc;
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return PHE_HIP_OK;
}

// ---- RCCL all-gather of ciphertext shards (include/phe_hip.h "multi-GPU") ---------------------------------------------
// RCCL is resolved with dlopen on first use: the library has no link-time dependency on it, single-GPU hosts never load it,
// and inside a torch process the already loaded librccl.so.1 (same SONAME) is the one that answers.
namespace {
struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, phe_rccl_id, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;
int rccl_load() {
    if (g_rccl.handle) return PHE_HIP_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* nm : names)
        if ((h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return fail(PHE_HIP_EHIP, std::string("RCCL not available: ") + dlerror());
    RcclApi a;
    a.handle = h;
    a.GetUniqueId = (int (*)(void*))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (int (*)(void**, int, phe_rccl_id, int))dlsym(h, "ncclCommInitRank");
    a.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(h, "ncclAllGather");
    a.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
    a.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.AllGather || !a.CommDestroy) return fail(PHE_HIP_EHIP, "RCCL symbols missing");
    g_rccl = a;
    return PHE_HIP_OK;
}
int rccl_fail(const char* what, int code) {
    return fail(PHE_HIP_EHIP, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(code) : "rccl error"));
}
}  // namespace

struct phe_hip_comm {
    void* comm = nullptr;
    int device = 0, rank = 0, world = 1;
};

int phe_hip_comm_unique_id(uint8_t id[128]) {
    if (!id) return fail(PHE_HIP_EINVAL, "null id");
    if (int rc = rccl_load()) return rc;
    phe_rccl_id uid;
    if (int e = g_rccl.GetUniqueId(&uid)) return rccl_fail("ncclGetUniqueId", e);
    memcpy(id, uid.internal, 128);
    return PHE_HIP_OK;
}

int phe_hip_comm_create(phe_hip_ctx* ctx, const uint8_t id[128], int rank, int world, phe_hip_comm** out) {
    if (check_ctx(ctx)) return PHE_HIP_EINVAL;
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return fail(PHE_HIP_EINVAL, "bad id / rank / world");
    if (int rc = rccl_load()) return rc;
    if (int rc = bind_device(ctx)) return rc;
    phe_rccl_id uid;
    memcpy(uid.internal, id, 128);
    phe_hip_comm* c = new phe_hip_comm();
    c->device = ctx->device;
    c->rank = rank;
    c->world = world;
    if (int e = g_rccl.CommInitRank(&c->comm, world, uid, rank)) {
        delete c;
        return rccl_fail("ncclCommInitRank", e);
    }
    *out = c;
    return PHE_HIP_OK;
}

int phe_hip_allgather_dev(phe_hip_comm* comm, const uint32_t* local, uint32_t* all, size_t rows, int limbs, void* stream) {
    if (!comm || !comm->comm) return fail(PHE_HIP_EINVAL, "null communicator");
    if (rows == 0) return PHE_HIP_OK;
    if (!local || !all || limbs < 1) return fail(PHE_HIP_EINVAL, "null buffer / limbs");
    HIP_TRY(hipSetDevice(comm->device));
    if (int e = g_rccl.AllGather(local, all, rows * (size_t)limbs, 3 /* ncclUint32 */, comm->comm, (hipStream_t)stream))
        return rccl_fail("ncclAllGather", e);
    return PHE_HIP_OK;
}

void phe_hip_comm_destroy(phe_hip_comm* comm) {
    if (!comm) return;
    if (comm->comm && g_rccl.CommDestroy) {
        (void)hipSetDevice(comm->device);
        (void)g_rccl.CommDestroy(comm->comm);
    }
    delete comm;
}

int phe_hip_selftest_prims(int device, uint32_t* out) {
    if (!out) return fail(PHE_HIP_EINVAL, "null out");
    HIP_TRY(hipSetDevice(device));
    uint32_t* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, 1026 * 4));
    k_selftest_prims<<<dim3(1), dim3(64), 0, nullptr>>>(d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d, 1026 * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipFree(d));
    return PHE_HIP_OK;
}

}  // extern "C"
