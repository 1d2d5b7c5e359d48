// tests/emu/emu_driver.cpp — TEST INFRASTRUCTURE.
// Compiles the device algorithm headers (csrc/mont_core.h, csrc/decrypt_tail.h) and the host
// key setup (csrc/key_setup.h) for the CPU, with tests/emu/wave_emu.h standing in for the
// wavefront, and exposes a C interface shaped like the product C-ABI so CPU-only tests can
// compare the very code the GPU runs against the oracle.  Not part of the product: the product
// library has no CPU execution path.
#include "wave_emu.h"
// clang-format off
#include "../../python-paillier_amd/csrc/mont_core.h"
#include "../../python-paillier_amd/csrc/mul_io.h"
#include "../../python-paillier_amd/csrc/split_core.h"
#include "../../python-paillier_amd/csrc/mul_table.h"
#include "../../python-paillier_amd/csrc/mul_tile.h"
#include "../../python-paillier_amd/csrc/decrypt_tail.h"
#include "../../python-paillier_amd/csrc/key_setup.h"
#include "../../python-paillier_amd/csrc/radix_conv.h"
#include "../../python-paillier_amd/csrc/primality.h"
// clang-format on
#include <string.h>

#include <string>
#include <vector>

using namespace phe;
using host::Big;

static thread_local std::string g_err;

#define DISPATCH_GL(G_, L_, CALL)                                                     \
    switch ((G_) * 100 + (L_)) {                                                      \
        case 6401: { constexpr int GG = 64, LL = 1; CALL; break; }                    \
        case 6402: { constexpr int GG = 64, LL = 2; CALL; break; }                    \
        case 6403: { constexpr int GG = 64, LL = 3; CALL; break; }                    \
        case 6405: { constexpr int GG = 64, LL = 5; CALL; break; }                    \
        case 1601: { constexpr int GG = 16, LL = 1; CALL; break; }                    \
        case 1602: { constexpr int GG = 16, LL = 2; CALL; break; }                    \
        case 1603: { constexpr int GG = 16, LL = 3; CALL; break; }                    \
        case 1604: { constexpr int GG = 16, LL = 4; CALL; break; }                    \
        case 1605: { constexpr int GG = 16, LL = 5; CALL; break; }                    \
        case 1607: { constexpr int GG = 16, LL = 7; CALL; break; }                    \
        case 1609: { constexpr int GG = 16, LL = 9; CALL; break; }                    \
        case 1614: { constexpr int GG = 16, LL = 14; CALL; break; }                   \
        case 1618: { constexpr int GG = 16, LL = 18; CALL; break; }                   \
        case 218: { constexpr int GG = 2, LL = 18; CALL; break; }                     \
        case 236: { constexpr int GG = 2, LL = 36; CALL; break; }                     \
        case 409: { constexpr int GG = 4, LL = 9; CALL; break; }                      \
        case 427: { constexpr int GG = 4, LL = 27; CALL; break; }                     \
        case 418: { constexpr int GG = 4, LL = 18; CALL; break; }                     \
        case 436: { constexpr int GG = 4, LL = 36; CALL; break; }                     \
        case 827: { constexpr int GG = 8, LL = 27; CALL; break; }                     \
        case 805: { constexpr int GG = 8, LL = 5; CALL; break; }                      \
        case 809: { constexpr int GG = 8, LL = 9; CALL; break; }                      \
        case 814: { constexpr int GG = 8, LL = 14; CALL; break; }                     \
        case 818: { constexpr int GG = 8, LL = 18; CALL; break; }                     \
        default: throw std::invalid_argument("unsupported geometry");                 \
    }

static ModConsts consts_of(const host::ModulusPack& m) {
    ModConsts c;
    c.n = m.n.data(); c.r1 = m.r1.data(); c.r2 = m.r2.data(); c.r3 = m.r3.data(); c.aux = m.aux.data();
    c.n0inv = m.n0inv;
    return c;
}

// number of emulated waves: enough groups for the batch, capped (groups stride over items like the GPU grid)
static int waves_for(uint64_t B, int G, int cap = 2) {
    const int per_wave = 64 / G;
    int w = (int)((B + per_wave - 1) / per_wave);
    return w < 1 ? 1 : (w > cap ? cap : w);
}

template <int G, int L, int MODE>
static void run_uniform(UniformArgs A) {
    constexpr int S = G * L, kPer = 64 / G;
    const int n_waves = waves_for(A.batch, G);
    const uint32_t total = (uint32_t)(kPer * n_waves);
    std::vector<uint32_t> table((size_t)total * (size_t)A.tbl_entries * S);
    A.table = table.data();
    for (int w = 0; w < n_waves; ++w) {
        std::vector<uint32_t> lds(kPer * (S + kLdsPad));
        wave::run_wave([&](uint32_t lane) {
            const uint32_t grp = lane / G;
            modexp_uniform_body<G, L, MODE>(A, lds.data() + grp * (S + kLdsPad), (uint32_t)w * kPer + grp, total, lane);
        });
    }
}

template <int G, int L>
static void run_var(VarArgs A) {
    constexpr int S = G * L, kPer = 64 / G;
    const int n_waves = waves_for(A.batch, G);
    const uint32_t total = (uint32_t)(kPer * n_waves);
    std::vector<uint32_t> table((size_t)total * ((size_t)1 << A.window) * S);
    A.table = table.data();
    for (int w = 0; w < n_waves; ++w) {
        std::vector<uint32_t> lds(kPer * (S + kLdsPad));
        wave::run_wave([&](uint32_t lane) {
            const uint32_t grp = lane / G;
            modexp_var_body<G, L>(A, lds.data() + grp * (S + kLdsPad), (uint32_t)w * kPer + grp, total, lane);
        });
    }
}

static int g_mul_io = 1;  // 1: mul_io.h (the kernels' body), 0: mont_core.h:mulmod_body (the reference form)

static bool rows_vec_ok(const MulArgs& A) {
    auto al = [](const void* p) { return ((uintptr_t)p & 15u) == 0; };
    return al(A.a) && al(A.b) && al(A.out) && A.a_stride % 4 == 0 && A.b_stride % 4 == 0 && A.out_stride % 4 == 0 &&
           A.limbs % 4 == 0 && A.b_plain_limbs % 4 == 0 && A.a_limbs % 4 == 0;
}

// the element-wise product the way k_mulmod runs it: the staged body (mul_io.h) where the geometry offers it and the rows
// are aligned, the plain one otherwise; returns which one ran
static int g_last_mul_staged = 0;
template <int G, int L>
static void run_mul(MulArgs A) {
    using IO = RowIO<G, L>;
    constexpr int S = G * L, kPer = 64 / G;
    const int n_waves = waves_for(A.batch, G);
    const uint32_t total = (uint32_t)(kPer * n_waves);
    A.vec_ok = rows_vec_ok(A) ? 1 : 0;
    const bool staged = g_mul_io && IO::kUse && A.vec_ok;
    g_last_mul_staged = staged ? 1 : 0;
    for (int w = 0; w < n_waves; ++w) {
        if (staged) {
            // one wave: its groups' digit rows | its staging area (a | b) | R^2, 16-byte aligned like the kernels' LDS
            std::vector<Words4> lds(((size_t)kPer * IO::kRow + 2 * IO::kStageWave + IO::kConstWords) / 4 + 1);
            uint32_t* base = (uint32_t*)lds.data();
            for (size_t i = 0; i < lds.size() * 4; ++i) base[i] = 0xdeadbeefu;   // LDS is not zero on the device either
            uint32_t* r2_row = base + kPer * IO::kRow + 2 * IO::kStageWave;
            for (int k = 0; k < S; ++k) r2_row[k] = A.mod.r2[k];
            wave::run_wave([&](uint32_t lane) {
                const uint32_t grp = lane / G;
                if (A.one_product)
                    mul_io_body<G, L, true>(A, base + grp * IO::kRow, base + kPer * IO::kRow, r2_row, (uint32_t)w * kPer + grp, total, lane);
                else
                    mul_io_body<G, L, false>(A, base + grp * IO::kRow, base + kPer * IO::kRow, r2_row, (uint32_t)w * kPer + grp, total, lane);
            });
        } else {
            std::vector<uint32_t> lds(kPer * (S + kLdsPad));
            wave::run_wave([&](uint32_t lane) {
                const uint32_t grp = lane / G;
                if (A.one_product)
                    mulmod_body<G, L, true>(A, lds.data() + grp * (S + kLdsPad), (uint32_t)w * kPer + grp, total, lane);
                else
                    mulmod_body<G, L, false>(A, lds.data() + grp * (S + kLdsPad), (uint32_t)w * kPer + grp, total, lane);
            });
        }
    }
}

// geometries of the split-modulus kernels (key_setup.h: kS2 / kS4 / kS8 / kS16)
#define DISPATCH_SPLIT(G_, L_, CALL)                                                  \
    switch ((G_) * 100 + (L_)) {                                                      \
        case 6401: { constexpr int GG = 64, LL = 1; CALL; break; }                    \
        case 6402: { constexpr int GG = 64, LL = 2; CALL; break; }                    \
        case 6403: { constexpr int GG = 64, LL = 3; CALL; break; }                    \
        case 6405: { constexpr int GG = 64, LL = 5; CALL; break; }                    \
        case 1601: { constexpr int GG = 16, LL = 1; CALL; break; }                    \
        case 1602: { constexpr int GG = 16, LL = 2; CALL; break; }                    \
        case 1603: { constexpr int GG = 16, LL = 3; CALL; break; }                    \
        case 1604: { constexpr int GG = 16, LL = 4; CALL; break; }                    \
        case 1605: { constexpr int GG = 16, LL = 5; CALL; break; }                    \
        case 1607: { constexpr int GG = 16, LL = 7; CALL; break; }                    \
        case 1609: { constexpr int GG = 16, LL = 9; CALL; break; }                    \
        case 1614: { constexpr int GG = 16, LL = 14; CALL; break; }                   \
        case 1618: { constexpr int GG = 16, LL = 18; CALL; break; }                   \
        case 118: { constexpr int GG = 1, LL = 18; CALL; break; }                     \
        case 209: { constexpr int GG = 2, LL = 9; CALL; break; }                      \
        case 218: { constexpr int GG = 2, LL = 18; CALL; break; }                     \
        case 227: { constexpr int GG = 2, LL = 27; CALL; break; }                     \
        case 427: { constexpr int GG = 4, LL = 27; CALL; break; }                     \
        case 405: { constexpr int GG = 4, LL = 5; CALL; break; }                      \
        case 803: { constexpr int GG = 8, LL = 3; CALL; break; }                      \
        case 409: { constexpr int GG = 4, LL = 9; CALL; break; }                      \
        case 414: { constexpr int GG = 4, LL = 14; CALL; break; }                     \
        case 418: { constexpr int GG = 4, LL = 18; CALL; break; }                     \
        case 805: { constexpr int GG = 8, LL = 5; CALL; break; }                      \
        case 807: { constexpr int GG = 8, LL = 7; CALL; break; }                      \
        case 809: { constexpr int GG = 8, LL = 9; CALL; break; }                      \
        case 814: { constexpr int GG = 8, LL = 14; CALL; break; }                     \
        case 818: { constexpr int GG = 8, LL = 18; CALL; break; }                     \
        default: throw std::invalid_argument("unsupported split geometry");           \
    }

static SplitConsts split_consts_of(const host::SplitPack& m) {
    SplitConsts c;
    c.nbar = c.kx = nullptr;
    c.n = m.n.data(); c.r1 = m.r1.data();
    c.e = m.e.data(); c.conv = m.conv.data(); c.nsq = m.nsq.data(); c.n0inv = m.n0inv; c.rows = m.rows;
    return c;
}
// the wave-pair kernels' two constant sets (key_setup.h QuickPack): the scaled modulus into A.mod, the true one into A.exit_mod
static void quick_consts_into(SplitArgs& A, const host::QuickPack& Q) {
    A.mod = split_consts_of(Q.scaled);
    A.mod.nbar = Q.nbar.data();
    A.exit_mod = split_consts_of(Q.exit);
    A.exit_mod.kx = Q.kx.data();
}
static int chunks_for(int limbs32, int H) { return std::max(1, (32 * limbs32 + 29 * H - 1) / (29 * H)); }

template <int G, int L, int MODE, bool U = false>
static void run_split(SplitArgs A) {
    constexpr int S2 = 2 * G * L, kPer = 64 / G;
    const int n_waves = waves_for(A.batch, G);
    const uint32_t total = (uint32_t)(kPer * n_waves);
    std::vector<uint32_t> table((size_t)total * (size_t)A.tbl_entries * S2);
    A.table = table.data();
    for (int w = 0; w < n_waves; ++w) {
        std::vector<uint32_t> lds(kPer * (S2 + kLdsPad));
        wave::run_wave([&](uint32_t lane) {
            const uint32_t grp = lane / G;
            modexp_split_body<G, L, MODE, U>(A, lds.data() + grp * (S2 + kLdsPad), (uint32_t)w * kPer + grp, total, lane);
        });
    }
}

// k_modexp_split_late: the single-wave kernels of the small-batch rungs on the late sweeps (split_core.h modexp_split_late_body)
template <int G, int L, int MODE>
static void run_split_late(SplitArgs A) {
    constexpr int S2 = 2 * G * L, kPer = 64 / G;
    if constexpr (G == 64 && L <= 9) {
        const int n_waves = waves_for(A.batch, G);
        const uint32_t total = (uint32_t)(kPer * n_waves);
        std::vector<uint32_t> table((size_t)total * (size_t)A.tbl_entries * S2);
        A.table = table.data();
        for (int w = 0; w < n_waves; ++w) {
            std::vector<uint32_t> lds(kPer * (S2 + kLdsPad), 0xdeadbeefu);   // LDS is not zero on the device either
            wave::run_wave([&](uint32_t lane) {
                const uint32_t grp = lane / G;
                modexp_split_late_body<G, L, MODE>(A, lds.data() + grp * (S2 + kLdsPad), (uint32_t)w * kPer + grp, total, lane);
            });
        }
    } else {
        throw std::invalid_argument("no late kernel for this geometry");
    }
}

// k_modexp_split_ab: one number on a PAIR of waves (role 0 = first words, 1 = second words), the two waves in two host
// threads joined by wave::block_barrier; one table of (tbl_entries + 1) pairs per number.  G = 64 only.
template <int L, int MODE>
static void run_split_ab(SplitArgs A) {
    constexpr int H = 64 * L;
    std::vector<uint32_t> table((size_t)A.batch * (size_t)(A.tbl_entries + 1) * 2 * H);
    for (uint64_t item = 0; item < A.batch; ++item) {
        std::vector<uint32_t> lds(ab_lds_words<L>(), 0xdeadbeefu);   // LDS is not zero on the device either
        uint32_t* tbl = table.data() + (size_t)item * (size_t)(A.tbl_entries + 1) * 2 * H;
        wave::run_block(2, [&](uint32_t role, uint32_t lane) {
            modexp_split_ab_body<L, MODE>(A, lds.data(), tbl, A.sched, item, true, role, lane);
        });
    }
}
#define DISPATCH_AB(L_, CALL)                                                         \
    switch (L_) {                                                                     \
        case 1: { constexpr int LL = 1; CALL; break; }                                \
        case 2: { constexpr int LL = 2; CALL; break; }                                \
        case 3: { constexpr int LL = 3; CALL; break; }                                \
        case 5: { constexpr int LL = 5; CALL; break; }                                \
        default: throw std::invalid_argument("no wave-pair kernel for this L");       \
    }
static int g_wave_tail = 0;   // 1: the CRT tail of a decrypt runs one ciphertext per wavefront (the library's choice for small batches)
static int g_wave_pairs = 0;  // 1: whole-wave geometry runs every exponentiation on a wave pair (the library's choice for a handful of numbers)
static int g_late = 0;        // 1: rungs of 16 lanes / the whole wave run encrypt and the decrypt halves on the late sweeps (the library's choice for small batches)

template <int G, int L, bool PAIR = false>
static void run_var_split(SplitVarArgs A) {
    constexpr int S2 = 2 * G * L, kPer = 64 / G;
    const int n_waves = waves_for(A.batch, G);
    const uint32_t total = (uint32_t)(kPer * n_waves);
    std::vector<uint32_t> table((size_t)total * ((size_t)1 << A.window) * S2);
    A.table = table.data();
    for (int w = 0; w < n_waves; ++w) {
        std::vector<uint32_t> lds(kPer * (S2 + kLdsPad));
        wave::run_wave([&](uint32_t lane) {
            const uint32_t grp = lane / G;
            modexp_var_split_body<G, L, PAIR>(A, lds.data() + grp * (S2 + kLdsPad), (uint32_t)w * kPer + grp, total, lane);
        });
    }
}

template <int G, int L>
static void run_multi_split(SplitMultiArgs A) {
    constexpr int S2 = 2 * G * L, kPer = 64 / G;
    const int n_waves = waves_for(A.n_chunks * A.n_row_blocks, G);
    const uint32_t total = (uint32_t)(kPer * n_waves);
    std::vector<uint32_t> table((size_t)total * (size_t)A.chunk * (((size_t)1 << A.window) - 1) * (A.base_inv ? 2 : 1) * S2);
    A.table = table.data();
    for (int w = 0; w < n_waves; ++w) {
        std::vector<uint32_t> lds(kPer * (S2 + kLdsPad));
        wave::run_wave([&](uint32_t lane) {
            const uint32_t grp = lane / G;
            multiexp_split_body<G, L>(A, lds.data() + grp * (S2 + kLdsPad), (uint32_t)w * kPer + grp, total, lane);
        });
    }
}

template <int G, int L>
static void run_multi_tables(SplitTableArgs A) {
    constexpr int S2 = 2 * G * L, kPer = 64 / G;
    const int n_waves = waves_for(A.batch * (A.base_inv ? 2 : 1), G);
    const uint32_t total = (uint32_t)(kPer * n_waves);
    for (int w = 0; w < n_waves; ++w) {
        std::vector<uint32_t> lds(kPer * (S2 + kLdsPad));
        wave::run_wave([&](uint32_t lane) {
            const uint32_t grp = lane / G;
            multiexp_tables_body<G, L>(A, lds.data() + grp * (S2 + kLdsPad), (uint32_t)w * kPer + grp, total, lane);
        });
    }
}
template <int G, int L>
static void run_multi_lookup(SplitLookupArgs A) {
    constexpr int S2 = 2 * G * L, kPer = 64 / G;
    const int n_waves = waves_for(A.rows, G);
    const uint32_t total = (uint32_t)(kPer * n_waves);
    for (int w = 0; w < n_waves; ++w) {
        std::vector<uint32_t> lds(kPer * (S2 + kLdsPad));
        wave::run_wave([&](uint32_t lane) {
            const uint32_t grp = lane / G;
            multiexp_lookup_body<G, L>(A, lds.data() + grp * (S2 + kLdsPad), (uint32_t)w * kPer + grp, total, lane);
        });
    }
}

template <int G, int L>
static void run_miller_rabin(MillerRabinArgs A) {
    constexpr int S = G * L, kPer = 64 / G;
    const int n_waves = waves_for(A.batch, G, 4);
    const uint32_t total = (uint32_t)(kPer * n_waves);
    for (int w = 0; w < n_waves; ++w) {
        std::vector<uint32_t> lds(kPer * (S + kLdsPad));
        wave::run_wave([&](uint32_t lane) {
            const uint32_t grp = lane / G;
            miller_rabin_body<G, L>(A, lds.data() + grp * (S + kLdsPad), (uint32_t)w * kPer + grp, total, lane);
        });
    }
}
static int g_prefer_group = 0;
static int g_unit = 1;    // 1: the scaled-modulus path for r^n where the key offers it (the library's large-batch form)
static int g_engine = 1;  // 1: split-modulus kernels where a geometry exists (the product default), 0: full-width only

// out = a*b mod n^2 (b_plain == 0) or a*(1 + n*m) mod n^2 (b = plaintexts of n_limbs words) on the PAIR form
// (split_core.h:mulmod_split_body): the product kernel of key widths without a full-width geometry.  rc 2: no split geometry
template <int G, int L>
static void run_mul_split(SplitMulArgs A) {
    constexpr int S2 = 2 * G * L, kPer = 64 / G;
    const int n_waves = waves_for(A.batch, G);
    const uint32_t total = (uint32_t)(kPer * n_waves);
    for (int w = 0; w < n_waves; ++w) {
        std::vector<uint32_t> lds(kPer * (S2 + kLdsPad));
        wave::run_wave([&](uint32_t lane) {
            const uint32_t grp = lane / G;
            mulmod_split_body<G, L>(A, lds.data() + grp * (S2 + kLdsPad), (uint32_t)w * kPer + grp, total, lane);
        });
    }
}

template <int G, int L>
static void run_crt_lift(CrtLiftArgs A) {
    constexpr int S2 = 2 * G * L, kPer = 64 / G;
    const int n_waves = waves_for(A.batch, G);
    const uint32_t total = (uint32_t)(kPer * n_waves);
    for (int w = 0; w < n_waves; ++w) {
        std::vector<uint32_t> lds(kPer * (S2 + kLdsPad));
        wave::run_wave([&](uint32_t lane) {
            const uint32_t grp = lane / G;
            crt_lift_body<G, L>(A, lds.data() + grp * (S2 + kLdsPad), (uint32_t)w * kPer + grp, total, lane);
        });
    }
}

// resident rows in the pair form (split_core.h): op 0 words -> pair, 1 pair -> words (* (1 + n*m) when plaintexts are given), 2 pair * pair
template <int G, int L>
static void run_pair(int op, PairArgs A) {
    constexpr int S2 = 2 * G * L, kPer = 64 / G;
    const int n_waves = waves_for(A.batch, G);
    const uint32_t total = (uint32_t)(kPer * n_waves);
    for (int w = 0; w < n_waves; ++w) {
        std::vector<uint32_t> lds(kPer * (S2 + kLdsPad));
        wave::run_wave([&](uint32_t lane) {
            const uint32_t grp = lane / G;
            uint32_t* row = lds.data() + grp * (S2 + kLdsPad);
            if (op == 0) to_pair_body<G, L>(A, row, (uint32_t)w * kPer + grp, total, lane);
            else if (op == 1) from_pair_body<G, L>(A, row, (uint32_t)w * kPer + grp, total, lane);
            else pair_mul_body<G, L>(A, row, (uint32_t)w * kPer + grp, total, lane);
        });
    }
}

// the full-width geometry the products / the CRT lift use for a modulus pack (phe_hip.hip:light_geometry)
static void light_geometry_of(const host::ModulusPack& M, int& G, int& L) {
    G = M.G; L = M.L;
    if (G == 4 && L == 36) { G = 8; L = 18; }
    else if (G == 2 && L == 36) { G = 4; L = 18; }
}

// k_mulmod_table (mul_table.h): a*b mod N as one plain product + one fold against the key's table; one emulated wave (4 limb
// groups) at a time with its own copy of the workgroup's LDS areas.  rc 2: not offered for this modulus (the table does not fit)
template <int L>
static void run_mul_table(TableMulArgs A, const host::TableMulPack& T) {
    constexpr int G = 16, S = G * L, kRowT = S + kTableRowSlack, kPer = 64 / G;
    using IO = RowIO<G, L>;
    const int n_waves = waves_for(A.batch, G, 8);
    const uint32_t total = (uint32_t)(kPer * n_waves);
    std::vector<Words4> tbl4(T.table.size() / 4 + 1), cst4(3 * S / 4 + 1);
    uint32_t* tbl = (uint32_t*)tbl4.data();
    uint32_t* cst = (uint32_t*)cst4.data();
    memcpy(tbl, T.table.data(), T.table.size() * 4);
    memcpy(cst, T.n.data(), S * 4);
    memcpy(cst + S, T.ncomp.data(), S * 4);
    memcpy(cst + 2 * S, T.ncomp1.data(), S * 4);
    for (int w = 0; w < n_waves; ++w) {
        std::vector<Words4> lds((size_t)(kPer * kRowT + 2 * IO::kStageWave) / 4 + 1);
        uint32_t* base = (uint32_t*)lds.data();
        for (size_t i = 0; i < lds.size() * 4; ++i) base[i] = 0xdeadbeefu;   // LDS is not zero on the device either
        wave::run_wave([&](uint32_t lane) {
            const uint32_t grp = lane / G;
            mul_table_body<L>(A, base + grp * kRowT, base + kPer * kRowT, tbl, cst, (uint32_t)w * kPer + grp, total, lane);
        });
    }
}
// k_mulmod_tile (mul_tile.h): the same product by tiles of 64 with the fold on lane = element; one emulated workgroup of 8
// waves (one host thread each, joined by the barriers) per block
template <int L, int W = kTileWaves>
static void run_mul_tile(TableMulArgs A, const host::TableMulPack& T, int n_blocks) {
    using TS = TileShape<L, W>;
    A.table = T.table_cols.data();
    A.digits_padded = T.digits_padded;
    A.tile_waves = T.tile_waves;
    static_assert(host::kTileWavesHost == kTileWaves, "the table's column blocks are the kernel's waves");
    if (T.tile_waves != W || T.S != TS::S) throw std::runtime_error("the pack is cut for another workgroup shape");
    if ((size_t)tile_lds_words<L, W>() != T.tile_lds_words) throw std::runtime_error("host and device disagree about the tile's LDS");
    for (int b = 0; b < n_blocks; ++b) {
        std::vector<Words4> lds((size_t)tile_lds_words<L, W>() / 4 + 1);
        uint32_t* tile = (uint32_t*)lds.data();
        for (size_t i = 0; i < lds.size() * 4; ++i) tile[i] = 0xdeadbeefu;   // LDS is not zero on the device either
        uint32_t* prod_carry = tile + TS::kRows * kTile;
        uint32_t* top = prod_carry + 2 * 2 * W * kTile;
        uint32_t* fold_carry = top + kTile * kTableRowSlack;
        uint32_t* cst = fold_carry + 2 * W * kTile;
        memcpy(cst, T.n.data(), TS::S * 4);
        memcpy(cst + TS::S, T.ncomp.data(), TS::S * 4);
        memcpy(cst + 2 * TS::S, T.ncomp1.data(), TS::S * 4);
        wave::run_block(W, [&](uint32_t wv, uint32_t lane) {
            mul_tile_body<L, W>(A, tile, prod_carry, top, fold_carry, cst, wv, (uint32_t)b, (uint32_t)n_blocks, lane);
        });
    }
}
static int g_tile_mul = 0, g_tile_blocks = 2;
extern "C" {

// e: 0 = mul_table.h (table in LDS), 1 = mul_tile.h on 16 waves, 2 = mul_tile.h on 8 waves (S = 8 L: the shape of 1024-bit keys)
void emu_set_tile_mul(int e, int blocks) { g_tile_mul = (e == 1 || e == 2) ? e : 0; g_tile_blocks = blocks > 0 ? blocks : 2; }
void emu_set_engine(int e) { g_engine = e ? 1 : 0; }
void emu_set_mul_io(int e) { g_mul_io = e ? 1 : 0; }
void emu_set_unit(int e) { g_unit = e ? 1 : 0; }
void emu_set_wave_pairs(int e) { g_wave_pairs = e ? 1 : 0; }
void emu_set_wave_tail(int e) { g_wave_tail = e ? 1 : 0; }
void emu_set_late(int e) { g_late = e ? 1 : 0; }
int emu_unit_offered(const uint32_t* n, int n_limbs) {
    try { return host::build_public(n, n_limbs, g_prefer_group).nunit.G ? 1 : 0; } catch (...) { return 0; }
}
int emu_last_mul_staged() { return g_last_mul_staged; }

// multiply-add lane-operations (wave::mad64 calls) since the last reset, over all 64 lanes of every emulated wave
uint64_t emu_mad_count(int reset) {
    const uint64_t v = wave::mad_counter();
    if (reset) wave::mad_counter() = 0;
    return v;
}

const char* emu_last_error() { return g_err.c_str(); }

void emu_set_group(int g) { g_prefer_group = (g == 1 || g == 2 || g == 4 || g == 8 || g == 16 || g == 64) ? g : 0; }

// 64/G independent products a[r]*b[r]*R^-1 (mod N), one per limb group.  All arrays hold 29-bit limbs,
// G*L words per number; a < R, b < 2N; the result is < 2N, almost-normalised (limbs < 2^29 + 2^8).
int emu_montmul(int G, int L, const uint32_t* a, const uint32_t* b, const uint32_t* n, uint32_t n0inv, uint32_t* out) {
    try {
        DISPATCH_GL(G, L, ({
            constexpr int S = GG * LL, kPer = 64 / GG;
            std::vector<uint32_t> lds(kPer * (S + kLdsPad));
            wave::run_wave([&](uint32_t lane) {
                const Lanes<GG> ln(lane);
                const uint32_t grp = lane / GG, g = ln.g;
                uint32_t* lrow = lds.data() + grp * (S + kLdsPad);
                uint32_t x[LL], y[LL], nn[LL], r[LL];
                load_row<LL>(x, a + grp * S, g);
                load_row<LL>(y, b + grp * S, g);
                load_row<LL>(nn, n, g);
                lds_put<LL>(lrow, x, g);
                montmul<GG, LL>(r, lrow, y, nn, n0inv, ln);
                store_row<LL>(out + grp * S, r, g);
            });
        }));
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// encrypt (c_in == nullptr) or obfuscate (c_in != nullptr: out = c_in * r^n mod n^2)
int emu_encrypt(const uint32_t* n, int n_limbs, const uint32_t* m, const uint32_t* r, const uint32_t* c_in,
                uint32_t* c_out, uint64_t B) {
    try {
        if (B == 0) return 0;
        host::PublicPlan P = host::build_public(n, n_limbs, g_prefer_group);
        if (g_engine && g_late && !(P.nsplit.G == 64 && g_wave_pairs)) {
            // the library's path for small batches on the rungs of 16 lanes / the whole wave: late sweeps, the plaintext factor
            // folded into the way out (obfuscate: the bare power, then the product kernel)
            if (!(P.nsplit.G >= 16 && P.nquick.ok() && P.nquick.scaled.L <= 9)) throw std::invalid_argument("no late kernel for this key / group");
            const host::QuickPack& Q = P.nquick;
            std::vector<uint32_t> power;
            SplitArgs A;
            memset(&A, 0, sizeof A);
            quick_consts_into(A, Q);
            A.sched = P.exp_n.ops.data(); A.n_ops = (int)P.exp_n.ops.size();
            A.first_idx = P.exp_n.first_idx; A.tbl_entries = P.exp_n.tbl_entries;
            A.base = r; A.base_limbs = P.s1; A.base_chunks = chunks_for(P.s1, Q.scaled.rows);
            A.post = c_in ? nullptr : m; A.post_limbs = P.s1; A.post_chunks = 1;
            if (c_in) power.resize((size_t)B * P.s2);
            A.out = c_in ? power.data() : c_out; A.out_limbs = P.s2; A.batch = B;
            if (Q.scaled.G == 64) { DISPATCH_SPLIT(Q.scaled.G, Q.scaled.L, (run_split_late<GG, LL, kModeEncrypt>(A))); }
            else { DISPATCH_SPLIT(Q.scaled.G, Q.scaled.L, (run_split<GG, LL, kModeEncrypt, true>(A))); }   // 16 lanes: the quick form
            if (c_in) {
                MulArgs Mu;
                memset(&Mu, 0, sizeof Mu);
                Mu.mod = consts_of(P.nsq); Mu.a = power.data(); Mu.a_stride = (size_t)P.s2;
                Mu.b = c_in; Mu.b_stride = (size_t)P.s2; Mu.out = c_out; Mu.out_stride = (size_t)P.s2; Mu.limbs = P.s2; Mu.batch = B;
                int MG, ML;
                light_geometry_of(P.nsq, MG, ML);
                DISPATCH_GL(MG, ML, (run_mul<GG, LL>(Mu)));
            }
            return 0;
        }
        if (g_engine && g_unit && P.nunit.G) {
            // the library's large-batch path: r^n modulo the scaled modulus n'^2 (quotient digits without a multiply), then
            // one pass of the product kernel takes that residue to (1 + n*m) * r^n mod n^2 (or c_in * r^n)
            const host::SplitPack& M = P.nunit;
            const int W = P.unit_words;
            std::vector<uint32_t> tmp((size_t)B * W + 4);
            uint32_t* t16 = tmp.data() + ((16 - ((uintptr_t)tmp.data() & 15u)) & 15u) / 4;   // 16-byte aligned rows
            SplitArgs A;
            memset(&A, 0, sizeof A);
            A.mod = split_consts_of(M);
            A.sched = P.exp_n.ops.data(); A.n_ops = (int)P.exp_n.ops.size();
            A.first_idx = P.exp_n.first_idx; A.tbl_entries = P.exp_n.tbl_entries;
            A.base = r; A.base_limbs = P.s1; A.base_chunks = chunks_for(P.s1, M.rows);
            A.post = nullptr; A.post_limbs = P.s1; A.post_chunks = 1;
            A.out = t16; A.out_limbs = W; A.batch = B;
            DISPATCH_SPLIT(M.G, M.L, (run_split<GG, LL, kModeEncrypt, true>(A)));
            MulArgs Mu;
            memset(&Mu, 0, sizeof Mu);
            Mu.mod = consts_of(P.nsq); Mu.a = t16; Mu.a_limbs = W; Mu.a_stride = (size_t)W;
            Mu.b = c_in ? c_in : m; Mu.b_stride = c_in ? (size_t)P.s2 : (size_t)P.s1; Mu.b_plain_limbs = c_in ? 0 : P.s1;
            Mu.out = c_out; Mu.out_stride = (size_t)P.s2; Mu.limbs = P.s2; Mu.batch = B;
            int MG, ML;
            light_geometry_of(P.nsq, MG, ML);
            DISPATCH_GL(MG, ML, (run_mul<GG, LL>(Mu)));
            return 0;
        }
        if (g_engine && P.nsplit.G) {
            const host::SplitPack& M = P.nsplit;
            SplitArgs A;
            memset(&A, 0, sizeof A);
            A.mod = split_consts_of(M);
            A.sched = P.exp_n.ops.data(); A.n_ops = (int)P.exp_n.ops.size();
            A.first_idx = P.exp_n.first_idx; A.tbl_entries = P.exp_n.tbl_entries;
            A.base = r; A.base_limbs = P.s1; A.base_chunks = chunks_for(P.s1, M.rows);
            A.post = c_in ? c_in : m; A.post_limbs = c_in ? P.s2 : P.s1; A.post_chunks = chunks_for(A.post_limbs, M.rows);
            A.out = c_out; A.out_limbs = P.s2; A.batch = B;
            if (!c_in && M.G == 64 && g_wave_pairs && !P.nquick.ok()) throw std::invalid_argument("no wave-pair constants for this key");
            if (!c_in && M.G == 64 && g_wave_pairs) {
                quick_consts_into(A, P.nquick);
                A.base_chunks = chunks_for(P.s1, P.nquick.scaled.rows);
                A.post_chunks = chunks_for(A.post_limbs, P.nquick.scaled.rows);
                DISPATCH_AB(P.nquick.scaled.L, (run_split_ab<LL, kModeEncrypt>(A)));
                return 0;
            }
            if (c_in) { DISPATCH_SPLIT(M.G, M.L, (run_split<GG, LL, kModeObfuscate>(A))); }
            else { DISPATCH_SPLIT(M.G, M.L, (run_split<GG, LL, kModeEncrypt>(A))); }
            return 0;
        }
        UniformArgs A;
        memset(&A, 0, sizeof A);
        A.mod = consts_of(P.nsq);
        A.sched = P.exp_n.ops.data(); A.n_ops = (int)P.exp_n.ops.size();
        A.first_idx = P.exp_n.first_idx; A.tbl_entries = P.exp_n.tbl_entries;
        A.base = r; A.base_limbs = P.s1;
        A.post = c_in ? c_in : m; A.post_limbs = c_in ? P.s2 : P.s1;
        A.out = c_out; A.out_limbs = P.s2; A.batch = B;
        if (c_in) { DISPATCH_GL(P.nsq.G, P.nsq.L, (run_uniform<GG, LL, kModeObfuscate>(A))); }
        else { DISPATCH_GL(P.nsq.G, P.nsq.L, (run_uniform<GG, LL, kModeEncrypt>(A))); }
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// raw_encrypt by the key owner (phe_hip_encrypt_owner_dev): r^n mod p^2 and mod q^2 by the half-exponentiation kernels
// with the exponent n, the CRT lift, then the product with 1 + n*m.  rc 2: not offered for this key (no geometry)
int emu_encrypt_owner(const uint32_t* n, const uint32_t* p, const uint32_t* q, const uint32_t* hp, const uint32_t* hq,
                      const uint32_t* p_inverse, int pq_limbs, int n_limbs, const uint32_t* m, const uint32_t* r,
                      uint32_t* c_out, uint64_t B) {
    try {
        if (B == 0) return 0;
        host::PublicPlan PUB = host::build_public(n, n_limbs, g_prefer_group);
        host::PrivatePlan P = host::build_private(p, q, hp, hq, p_inverse, pq_limbs, n_limbs, g_prefer_group);
        int G, L;
        light_geometry_of(P.qsq, G, L);
        if (!P.psplit.G || !P.qsplit.G || !PUB.nsq.G || !host::split_part_holds(G, L)) return 2;
        host::OwnerLift W;
        if (!host::build_owner_lift(P.tail.p, P.tail.q, P.qsq.S, W)) return 2;
        const int S = (std::max(P.psq.bits, P.qsq.bits) + 31) / 32;
        std::vector<uint32_t> yp((size_t)B * S), yq((size_t)B * S);
        for (int half = 0; half < 2; ++half) {
            const host::SplitPack& SP = half ? P.qsplit : P.psplit;
            SplitArgs A;
            memset(&A, 0, sizeof A);
            A.mod = split_consts_of(SP);
            A.sched = PUB.exp_n.ops.data(); A.n_ops = (int)PUB.exp_n.ops.size();
            A.first_idx = PUB.exp_n.first_idx; A.tbl_entries = PUB.exp_n.tbl_entries;
            A.base = r; A.base_limbs = P.s1; A.base_chunks = chunks_for(P.s1, SP.rows);
            A.out = half ? yq.data() : yp.data(); A.out_limbs = S; A.batch = B;
            const host::QuickPack& QP = half ? P.qquick : P.pquick;
            if (g_late && SP.G >= 16 && QP.ok() && QP.scaled.L <= 9) {
                quick_consts_into(A, QP);
                A.base_chunks = chunks_for(P.s1, QP.scaled.rows);
                if (QP.scaled.G == 64) { DISPATCH_SPLIT(QP.scaled.G, QP.scaled.L, (run_split_late<GG, LL, kModeHalfDecrypt>(A))); }
                else { DISPATCH_SPLIT(QP.scaled.G, QP.scaled.L, (run_split<GG, LL, kModeHalfDecrypt, true>(A))); }
                continue;
            }
            DISPATCH_SPLIT(SP.G, SP.L, (run_split<GG, LL, kModeHalfDecrypt>(A)));
        }
        CrtLiftArgs A;
        memset(&A, 0, sizeof A);
        A.mod = consts_of(P.qsq);
        A.kr = W.kr.data(); A.nkr = W.nkr.data(); A.psq = W.psq.data();
        A.yp = yp.data(); A.yq = yq.data(); A.x_stride = (size_t)S; A.x_limbs = S;
        A.out = c_out; A.out_limbs = P.s2; A.batch = B;
        DISPATCH_SPLIT(G, L, (run_crt_lift<GG, LL>(A)));
        MulArgs Mu;
        memset(&Mu, 0, sizeof Mu);
        Mu.mod = consts_of(PUB.nsq); Mu.a = c_out; Mu.b = m; Mu.out = c_out; Mu.limbs = PUB.s2; Mu.batch = B;
        Mu.a_stride = Mu.out_stride = (size_t)PUB.s2; Mu.b_stride = (size_t)PUB.s1; Mu.b_plain_limbs = PUB.s1;
        int MG, ML;
        light_geometry_of(PUB.nsq, MG, ML);
        DISPATCH_GL(MG, ML, (run_mul<GG, LL>(Mu)));
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

int emu_decrypt(const uint32_t* p, const uint32_t* q, const uint32_t* hp, const uint32_t* hq,
                const uint32_t* p_inverse, int pq_limbs, int n_limbs, const uint32_t* c, uint32_t* m_out,
                uint64_t B) {
    try {
        if (B == 0) return 0;
        host::PrivatePlan P = host::build_private(p, q, hp, hq, p_inverse, pq_limbs, n_limbs, g_prefer_group);
        const int S = (std::max(P.psq.bits, P.qsq.bits) + 31) / 32;
        std::vector<uint32_t> xp((size_t)B * S), xq((size_t)B * S);
        for (int half = 0; half < 2; ++half) {
            const host::ModulusPack& M = half ? P.qsq : P.psq;
            const host::Schedule& E = half ? P.exp_q : P.exp_p;
            const host::SplitPack& SP = half ? P.qsplit : P.psplit;
            if (g_engine && SP.G) {
                SplitArgs A;
                memset(&A, 0, sizeof A);
                A.mod = split_consts_of(SP);
                A.sched = E.ops.data(); A.n_ops = (int)E.ops.size();
                A.first_idx = E.first_idx; A.tbl_entries = E.tbl_entries;
                A.base = c; A.base_limbs = P.s2; A.base_chunks = chunks_for(P.s2, SP.rows);
                A.out = half ? xq.data() : xp.data(); A.out_limbs = S; A.batch = B;
                const host::QuickPack& QP = half ? P.qquick : P.pquick;
                if (SP.G == 64 && g_wave_pairs && !QP.ok()) throw std::invalid_argument("no wave-pair constants for this key");
                if (SP.G == 64 && g_wave_pairs) {
                    quick_consts_into(A, QP);
                    A.base_chunks = chunks_for(P.s2, QP.scaled.rows);
                    DISPATCH_AB(QP.scaled.L, (run_split_ab<LL, kModeHalfDecrypt>(A)));
                    continue;
                }
                if (g_late) {
                    if (!(SP.G >= 16 && QP.ok() && QP.scaled.L <= 9)) throw std::invalid_argument("no late kernel for this key / group");
                    quick_consts_into(A, QP);
                    A.base_chunks = chunks_for(P.s2, QP.scaled.rows);
                    if (QP.scaled.G == 64) { DISPATCH_SPLIT(QP.scaled.G, QP.scaled.L, (run_split_late<GG, LL, kModeHalfDecrypt>(A))); }
                    else { DISPATCH_SPLIT(QP.scaled.G, QP.scaled.L, (run_split<GG, LL, kModeHalfDecrypt, true>(A))); }
                    continue;
                }
                DISPATCH_SPLIT(SP.G, SP.L, (run_split<GG, LL, kModeHalfDecrypt>(A)));
                continue;
            }
            UniformArgs A;
            memset(&A, 0, sizeof A);
            A.mod = consts_of(M);
            A.sched = E.ops.data(); A.n_ops = (int)E.ops.size();
            A.first_idx = E.first_idx; A.tbl_entries = E.tbl_entries;
            A.base = c; A.base_limbs = P.s2;
            A.out = half ? xq.data() : xp.data(); A.out_limbs = S; A.batch = B;
            DISPATCH_GL(M.G, M.L, (run_uniform<GG, LL, kModeHalfDecrypt>(A)));
        }
        if (g_wave_tail) {
            // the library's tail for small batches: one ciphertext per wavefront (split_core.h decrypt_tail_wave_body)
            const host::TailWavePack TW = host::build_tail_wave(P.tail);
            if (!TW.ok()) throw std::invalid_argument("no wave tail for this key width");
            TailWaveArgs W;
            memset(&W, 0, sizeof W);
            W.k.p = TW.p.data(); W.k.q = TW.q.data(); W.k.pinv = TW.pinv.data(); W.k.qinv = TW.qinv.data();
            W.k.hp_r = TW.hp_r.data(); W.k.hq_r = TW.hq_r.data(); W.k.pinvq_r = TW.pinvq_r.data();
            W.k.p0inv = TW.p0inv; W.k.q0inv = TW.q0inv; W.k.rows = TW.rows;
            W.xp = xp.data(); W.xq = xq.data(); W.x_stride = S; W.m_out = m_out; W.out_limbs = P.s1; W.batch = B;
            for (uint64_t i = 0; i < B; ++i) {
                std::vector<uint32_t> lds(2 * 64 * TW.L + kLdsPad, 0xdeadbeefu);
                DISPATCH_AB(TW.L, (wave::run_wave([&](uint32_t lane) { decrypt_tail_wave_body<LL>(W, lds.data(), i, lane); })));
            }
            return 0;
        }
        TailArgs T;
        memset(&T, 0, sizeof T);
        const host::TailPack& K = P.tail;
        T.k.h = K.h; T.k.p = K.p.data(); T.k.q = K.q.data(); T.k.pinvw = K.pinvw.data(); T.k.qinvw = K.qinvw.data();
        T.k.hp_r = K.hp_r.data(); T.k.hq_r = K.hq_r.data(); T.k.pinvq_r = K.pinvq_r.data();
        T.k.p0inv = K.p0inv; T.k.q0inv = K.q0inv;
        T.xp = xp.data(); T.xq = xq.data(); T.x_stride = S; T.m_out = m_out; T.out_limbs = P.s1; T.batch = B;
        std::vector<uint32_t> wsbuf((size_t)tail_ws_words(K.h));
        for (uint64_t i = 0; i < B; ++i) {
            TailWs ws{wsbuf.data(), 1};
            decrypt_tail_one(T, ws, i);
        }
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// what the LIBRARY offers for phe_hip_mulmod on this modulus besides the two Montgomery products (phe_hip.hip launch_mul:
// build_table_mul without the narrow lane widths, the 8-wave tile shape first: ctx creation): bit 0 the table in LDS (mul_table.h),
// bit 1 tiles (mul_tile.h: large batches), bit 2 (with bit 1) those tiles are the 8-wave shape (S = 8 L: keys of ~810 ... 1025 bits)
int emu_table_mul_offered(const uint32_t* N, int limbs) {
    try {
        const host::TableMulPack T8 = host::build_table_mul(host::big_from(N, limbs, limbs), limbs, false, 8);
        if (T8.ok() && T8.tiles()) return 2 | 4;
        const host::TableMulPack T = host::build_table_mul(host::big_from(N, limbs, limbs), limbs, false);
        return !T.ok() ? 0 : ((T.in_lds() ? 1 : 0) | (T.tiles() ? 2 : 0));
    } catch (...) { return 0; }
}

int emu_mulmod_table(const uint32_t* N, int limbs, const uint32_t* a, const uint32_t* b, uint32_t* out, uint64_t B) {
    try {
        if (B == 0) return 0;
        const host::TableMulPack T = host::build_table_mul(host::big_from(N, limbs, limbs), limbs, true, g_tile_mul == 2 ? 8 : 16);
        if (!T.ok()) return 2;
        std::vector<Words4> a4((size_t)B * limbs / 4 + 1), b4((size_t)B * limbs / 4 + 1), o4((size_t)B * limbs / 4 + 1);   // 16-byte aligned rows
        memcpy(a4.data(), a, (size_t)B * limbs * 4);
        memcpy(b4.data(), b, (size_t)B * limbs * 4);
        TableMulArgs A;
        memset(&A, 0, sizeof A);
        A.n = T.n.data(); A.ncomp = T.ncomp.data(); A.ncomp1 = T.ncomp1.data(); A.table = T.table.data();
        A.inv = T.inv; A.split = T.split; A.digits = T.digits; A.base = T.base;
        A.a = (const uint32_t*)a4.data(); A.b = (const uint32_t*)b4.data(); A.out = (uint32_t*)o4.data();
        A.a_stride = A.b_stride = A.out_stride = (size_t)limbs; A.limbs = limbs; A.batch = B;
        if (limbs % 4) return 2;
        if (!g_tile_mul && !T.in_lds()) return 2;
        if (g_tile_mul) {
            if (!T.tiles()) return 2;
            if (g_tile_mul == 2) {
                if (T.L == 9) run_mul_tile<9, 8>(A, T, g_tile_blocks);
                else return 2;
            } else if (T.L == 5) run_mul_tile<5>(A, T, g_tile_blocks);
            else if (T.L == 9) run_mul_tile<9>(A, T, g_tile_blocks);
            else if (T.L == 14) run_mul_tile<14>(A, T, g_tile_blocks);
            else return 2;
        } else if (T.L == 5) run_mul_table<5>(A, T);
        else if (T.L == 9) run_mul_table<9>(A, T);
        else return 2;
        memcpy(out, o4.data(), (size_t)B * limbs * 4);
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// out = a*b mod N, rows of `limbs` words (N given in `limbs` words too)
int emu_mulmod(const uint32_t* N, int limbs, const uint32_t* a, const uint32_t* b, uint32_t* out, uint64_t B) {
    try {
        if (B == 0) return 0;
        host::ModulusPack M = host::build_modulus(host::big_from(N, limbs, limbs), nullptr, 32 * limbs, g_prefer_group);
        MulArgs A;
        memset(&A, 0, sizeof A);
        A.mod = consts_of(M); A.a = a; A.b = b; A.limbs = limbs; A.out = out; A.batch = B;
        A.a_stride = A.b_stride = A.out_stride = (size_t)limbs;
        DISPATCH_GL(M.G, M.L, (run_mul<GG, LL>(A)));
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

int emu_mulmod_n2_split(const uint32_t* n, int n_limbs, const uint32_t* a, const uint32_t* b, int b_plain, uint32_t* out,
                        uint64_t B) {
    try {
        if (B == 0) return 0;
        host::SplitPack M = host::build_split(host::big_from(n, n_limbs, n_limbs), 64 * n_limbs, g_prefer_group);
        if (M.G == 0) return 2;
        SplitMulArgs A;
        memset(&A, 0, sizeof A);
        A.mod = split_consts_of(M);
        A.a = a; A.b = b; A.out = out; A.limbs = 2 * n_limbs; A.chunks = chunks_for(2 * n_limbs, M.rows);
        A.a_stride = A.out_stride = (size_t)(2 * n_limbs);
        A.b_stride = b_plain ? (size_t)n_limbs : (size_t)(2 * n_limbs);
        A.b_plain_limbs = b_plain ? n_limbs : 0;
        A.batch = B;
        DISPATCH_SPLIT(M.G, M.L, (run_mul_split<GG, LL>(A)));
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// one Montgomery product per row: out = a*b*R^-1 mod N, canonical (R = 2^(29*S) of the geometry, returned through
// radix_bits); broadcast_b != 0: one row b for the whole batch
int emu_montmul_rows(const uint32_t* N, int limbs, const uint32_t* a, const uint32_t* b, int broadcast_b, uint32_t* out,
                     uint64_t B, int* radix_bits) {
    try {
        host::ModulusPack M = host::build_modulus(host::big_from(N, limbs, limbs), nullptr, 32 * limbs, g_prefer_group);
        if (radix_bits) *radix_bits = kRadixBits * M.S;
        if (B == 0) return 0;
        MulArgs A;
        memset(&A, 0, sizeof A);
        A.mod = consts_of(M); A.a = a; A.b = b; A.limbs = limbs; A.out = out; A.batch = B; A.one_product = 1;
        A.a_stride = A.out_stride = (size_t)limbs; A.b_stride = broadcast_b ? 0 : (size_t)limbs;
        DISPATCH_GL(M.G, M.L, (run_mul<GG, LL>(A)));
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// out = c * (1 + n*m) mod n^2 (the add-a-plaintext entry)
int emu_add_plain(const uint32_t* n, int n_limbs, const uint32_t* c, const uint32_t* m, uint32_t* out, uint64_t B) {
    try {
        if (B == 0) return 0;
        host::PublicPlan P = host::build_public(n, n_limbs, g_prefer_group);
        MulArgs A;
        memset(&A, 0, sizeof A);
        A.mod = consts_of(P.nsq); A.a = c; A.b = m; A.out = out; A.limbs = P.s2; A.batch = B;
        A.a_stride = A.out_stride = (size_t)P.s2; A.b_stride = (size_t)P.s1; A.b_plain_limbs = P.s1;
        DISPATCH_GL(P.nsq.G, P.nsq.L, (run_mul<GG, LL>(A)));
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// out = base^exp mod N with per-row exponents
int emu_powmod_var(const uint32_t* N, int limbs, const uint32_t* base, const uint32_t* exps, int exp_limbs,
                   uint32_t* out, uint64_t B) {
    try {
        if (B == 0) return 0;
        host::ModulusPack M = host::build_modulus(host::big_from(N, limbs, limbs), nullptr, 32 * limbs, g_prefer_group);
        int max_bits = 0;
        for (uint64_t i = 0; i < B; ++i) {
            Big e = host::big_from(exps + i * exp_limbs, exp_limbs, exp_limbs);
            max_bits = std::max(max_bits, host::big_bits(e));
        }
        VarArgs A;
        memset(&A, 0, sizeof A);
        A.mod = consts_of(M); A.base = base; A.base_limbs = limbs; A.exps = exps; A.exp_limbs = exp_limbs;
        A.window = host::pick_window(max_bits);
        A.n_windows = std::max(1, (max_bits + A.window - 1) / A.window);
        A.out = out; A.out_limbs = limbs; A.batch = B;
        DISPATCH_GL(M.G, M.L, (run_var<GG, LL>(A)));
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// Unit access to the pair arithmetic of split_core.h for 64/G independent groups: X, Y are pairs given as 2H 29-bit
// limbs (word 0 | word 1, any lazily reduced values below R); op 0: Z = X*Y, op 1: Z = X^2 (Z written as 2H limbs),
// op 2: the canonical residue of value(X) as `out_limbs` 32-bit words per group.
int emu_split_pair_op(int G, int L, const uint32_t* n, int n_limbs, int op, const uint32_t* X, const uint32_t* Y,
                      uint32_t* out, int out_limbs) {
    try {
        host::SplitPack M = host::build_split(host::big_from(n, n_limbs, n_limbs), 64 * n_limbs, 0);
        // constants are rows of H limbs in global limb order: any (G, L) with the same H can use them
        if (G * L != M.H) throw std::invalid_argument("G*L must equal the H key_setup picks for this modulus");
        const SplitConsts C = split_consts_of(M);
        DISPATCH_SPLIT(G, L, ({
            constexpr int H = GG * LL, S2 = 2 * H, kPer = 64 / GG;
            std::vector<uint32_t> lds(kPer * (S2 + kLdsPad));
            wave::run_wave([&](uint32_t lane) {
                const Lanes<GG> ln(lane);
                const uint32_t grp = lane / GG, g = ln.g;
                SplitLane<GG, LL> K;
                load_row<LL>(K.n, C.n, g);
                K.n0inv = C.n0inv;
                K.rows_ = C.rows;
                K.row_a = lds.data() + grp * (S2 + kLdsPad);
                K.row_c = K.row_a + H;
                uint32_t x0[LL], x1[LL], y0[LL], y1[LL];
                load_row<LL>(x0, X + grp * S2, g);
                load_row<LL>(x1, X + grp * S2 + H, g);
                if (op == 0) {
                    load_row<LL>(y0, Y + grp * S2, g);
                    load_row<LL>(y1, Y + grp * S2 + H, g);
                    split_mul<GG, LL>(x0, x1, y0, y1, K, ln);
                } else if (op == 1) {
                    split_square<GG, LL>(x0, x1, K, ln);
                }
                if (op == 2) {
                    split_exit<GG, LL>(out + grp * out_limbs, out_limbs, x0, x1, nullptr, 0, C, K, ln, true);
                } else {
                    store_row<LL>(out + grp * S2, x0, g);
                    store_row<LL>(out + grp * S2 + H, x1, g);
                }
            });
        }));
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// the pair-form entry points (phe_hip_to_pair_dev / from_pair_dev / pair_mul_dev).  The rows are those of the geometry that
// build_public picks WITHOUT a group preference (rung 0 of the library); `group` > 0 runs them on a wider geometry with the
// same H, as the library's ladder does for small batches.  *pair_words = 2H.  rc 2: no split geometry / no such rung.
int emu_pair_op(const uint32_t* n, int n_limbs, int op, int group, const uint32_t* a, const uint32_t* b, int b_is_row,
                uint32_t* out, uint64_t B, int* pair_words) {
    try {
        const host::PublicPlan P0 = host::build_public(n, n_limbs, 0);
        if (!P0.nsplit.G) return 2;
        if (pair_words) *pair_words = 2 * P0.nsplit.H;
        if (B == 0) return 0;
        host::SplitPack M = P0.nsplit;
        if (group > 0) {
            M = host::build_public(n, n_limbs, group).nsplit;
            if (M.G == 0 || M.H != P0.nsplit.H || M.rows != M.H) return 2;
        }
        PairArgs A;
        memset(&A, 0, sizeof A);
        A.mod = split_consts_of(M);
        A.a = a; A.b = b; A.out = out;
        A.limbs = P0.s2; A.chunks = chunks_for(P0.s2, M.rows);
        A.b_stride = (op == 2 && !b_is_row) ? (size_t)(2 * M.H) : 0;
        A.b_limbs = (op == 1 && b) ? P0.s1 : 0;
        A.batch = B;
        DISPATCH_SPLIT(M.G, M.L, (run_pair<GG, LL>(op, A)));
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// out[i] = a[i]^exps[i] on pair-form rows (phe_hip_pair_powmod_dev); group as in emu_pair_op
int emu_pair_powmod(const uint32_t* n, int n_limbs, int group, const uint32_t* a, const uint32_t* exps, int exp_limbs, uint32_t* out,
                    uint64_t B) {
    try {
        const host::PublicPlan P0 = host::build_public(n, n_limbs, 0);
        if (!P0.nsplit.G) return 2;
        if (B == 0) return 0;
        host::SplitPack M = P0.nsplit;
        if (group > 0) {
            M = host::build_public(n, n_limbs, group).nsplit;
            if (M.G == 0 || M.G == 64 || M.H != P0.nsplit.H || M.rows != M.H) return 2;
        }
        int max_bits = 1;
        for (uint64_t i = 0; i < B; ++i)
            max_bits = std::max(max_bits, host::big_bits(host::big_from(exps + i * exp_limbs, exp_limbs, exp_limbs)));
        SplitVarArgs A;
        memset(&A, 0, sizeof A);
        A.pair_io = 1;
        A.mod = split_consts_of(M);
        A.base = a;
        A.exps = exps; A.exp_limbs = exp_limbs;
        A.window = host::pick_window(max_bits);
        A.n_windows = std::max(1, (max_bits + A.window - 1) / A.window);
        A.out = out; A.batch = B;
        DISPATCH_SPLIT(M.G, M.L, (run_var_split<GG, LL, true>(A)));
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// out = base^exp mod n^2 with per-row exponents, the way phe_hip_powmod runs it (split-modulus kernel when the
// engine is on and a geometry exists, else the full-width kernel on n^2)
int emu_powmod_n2(const uint32_t* n, int n_limbs, const uint32_t* base, const uint32_t* exps, int exp_limbs,
                  uint32_t* out, uint64_t B) {
    try {
        if (B == 0) return 0;
        host::PublicPlan P = host::build_public(n, n_limbs, g_prefer_group);
        int max_bits = 0;
        for (uint64_t i = 0; i < B; ++i)
            max_bits = std::max(max_bits, host::big_bits(host::big_from(exps + i * exp_limbs, exp_limbs, exp_limbs)));
        if (g_engine && P.nsplit.G == 64 && g_wave_pairs && P.nquick.ok()) {
            // phe_hip_powmod for a handful of numbers: every number on a wave pair with its OWN sliding-window schedule (the
            // kernel wrapper's SplitArgs::item_meta); a zero exponent keeps the general path
            bool all_positive = true;
            for (uint64_t i = 0; i < B; ++i)
                all_positive = all_positive && host::big_bits(host::big_from(exps + i * exp_limbs, exp_limbs, exp_limbs)) > 0;
            if (all_positive) {
                for (uint64_t i = 0; i < B; ++i) {
                    const host::Schedule S = host::build_schedule(host::big_from(exps + i * exp_limbs, exp_limbs, exp_limbs));
                    SplitArgs A;
                    memset(&A, 0, sizeof A);
                    quick_consts_into(A, P.nquick);
                    A.sched = S.ops.data(); A.n_ops = (int)S.ops.size();
                    A.first_idx = S.first_idx; A.tbl_entries = S.tbl_entries;
                    A.base = base + i * (uint64_t)P.s2; A.base_limbs = P.s2; A.base_chunks = chunks_for(P.s2, P.nquick.scaled.rows);
                    A.post = nullptr; A.post_limbs = P.s1; A.post_chunks = 1;
                    A.out = out + i * (uint64_t)P.s2; A.out_limbs = P.s2; A.batch = 1;
                    DISPATCH_AB(P.nquick.scaled.L, (run_split_ab<LL, kModeEncrypt>(A)));
                }
                return 0;
            }
        }
        if (g_engine && P.nsplit.G) {
            const host::SplitPack& M = P.nsplit;
            SplitVarArgs A;
            memset(&A, 0, sizeof A);
            A.mod = split_consts_of(M);
            A.base = base; A.base_limbs = P.s2; A.base_chunks = chunks_for(P.s2, M.rows);
            A.exps = exps; A.exp_limbs = exp_limbs;
            A.window = host::pick_window(max_bits);
            A.n_windows = std::max(1, (max_bits + A.window - 1) / A.window);
            A.out = out; A.out_limbs = P.s2; A.batch = B;
            DISPATCH_SPLIT(M.G, M.L, (run_var_split<GG, LL>(A)));
            return 0;
        }
        VarArgs A;
        memset(&A, 0, sizeof A);
        A.mod = consts_of(P.nsq); A.base = base; A.base_limbs = P.s2; A.exps = exps; A.exp_limbs = exp_limbs;
        A.window = host::pick_window(max_bits);
        A.n_windows = std::max(1, (max_bits + A.window - 1) / A.window);
        A.out = out; A.out_limbs = P.s2; A.batch = B;
        DISPATCH_GL(P.nsq.G, P.nsq.L, (run_var<GG, LL>(A)));
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// k_multiexp_split alone: out[(j * rows + r)] = prod over the chunk [j*chunk, (j+1)*chunk) of b[i]^exps[r][i], b = base or
// base_inv where neg[r][i] (base_inv / neg may be null).  Returns 2 when the key has no split geometry (the product
// then takes powmod + the mulmod tree).
static int g_multi_pair_in = 0;  // 1: emu_multiexp_n2's bases are rows in the pair form (2H limbs each)
void emu_set_multiexp_pair_in(int e) { g_multi_pair_in = e ? 1 : 0; }
int emu_multiexp_n2(const uint32_t* n, int n_limbs, const uint32_t* base, const uint32_t* base_inv, const uint32_t* exps,
                    const uint8_t* neg, int exp_limbs, int chunk, int row_block, uint32_t* out, uint64_t rows, uint64_t B) {
    try {
        if (B == 0 || rows == 0 || chunk < 1 || row_block < 1 || (neg && !base_inv)) throw std::invalid_argument("bad multiexp shape");
        host::PublicPlan P = host::build_public(n, n_limbs, g_prefer_group);
        if (!P.nsplit.G) return 2;
        int max_bits = 1;
        for (uint64_t i = 0; i < B * rows; ++i)
            max_bits = std::max(max_bits, host::big_bits(host::big_from(exps + i * exp_limbs, exp_limbs, exp_limbs)));
        const host::SplitPack& M = P.nsplit;
        SplitMultiArgs A;
        memset(&A, 0, sizeof A);
        A.mod = split_consts_of(M);
        A.base = base; A.base_inv = base_inv; A.base_limbs = P.s2; A.base_chunks = chunks_for(P.s2, M.rows);
        if (g_multi_pair_in) { A.pair_in = 1; A.base_limbs = 2 * M.H; }
        A.exps = exps; A.neg = neg; A.exp_limbs = exp_limbs;
        A.window = host::pick_multi_window(max_bits);
        A.n_windows = std::max(1, (max_bits + A.window - 1) / A.window);
        A.chunk = chunk; A.row_block = row_block;
        A.out = out; A.out_limbs = P.s2; A.batch = B; A.rows = rows;
        A.n_chunks = (B + chunk - 1) / chunk;
        A.n_row_blocks = (rows + row_block - 1) / row_block;
        DISPATCH_SPLIT(M.G, M.L, (run_multi_split<GG, LL>(A)));
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// k_multiexp_tables + k_multiexp_lookup: out[r] = prod over the entries of row r of b[col]^exp, b = base or base_inv where
// neg[entry]; row_ptr/cols null = dense rows of B entries.  Returns 2 without a split geometry.
int emu_multiexp_csr(const uint32_t* n, int n_limbs, const uint32_t* base, const uint32_t* base_inv, uint64_t B,
                     const uint64_t* row_ptr, const uint32_t* cols, const uint32_t* exps, const uint8_t* neg, int exp_limbs,
                     uint64_t entries, const uint32_t* order, uint32_t* out, uint64_t rows) {
    try {
        if (B == 0 || rows == 0 || (neg && !base_inv)) throw std::invalid_argument("bad multiexp shape");
        host::PublicPlan P = host::build_public(n, n_limbs, g_prefer_group);
        if (!P.nsplit.G) return 2;
        int max_bits = 1;
        for (uint64_t i = 0; i < entries; ++i)
            max_bits = std::max(max_bits, host::big_bits(host::big_from(exps + i * exp_limbs, exp_limbs, exp_limbs)));
        const host::SplitPack& M = P.nsplit;
        const int w = host::pick_multi_window(max_bits);
        const int signs = base_inv ? 2 : 1;
        std::vector<uint32_t> table((size_t)B * signs * (((size_t)1 << w) - 1) * 2 * M.H);
        SplitTableArgs T;
        memset(&T, 0, sizeof T);
        T.mod = split_consts_of(M);
        T.base = base; T.base_inv = base_inv; T.base_limbs = P.s2; T.base_chunks = chunks_for(P.s2, M.rows);
        T.window = w; T.table = table.data(); T.batch = B;
        DISPATCH_SPLIT(M.G, M.L, (run_multi_tables<GG, LL>(T)));
        SplitLookupArgs A;
        memset(&A, 0, sizeof A);
        A.mod = T.mod; A.table = table.data(); A.signs = signs;
        A.row_ptr = row_ptr; A.cols = cols; A.exps = exps; A.neg = neg; A.order = order;
        A.exp_limbs = exp_limbs; A.window = w; A.n_windows = std::max(1, (max_bits + w - 1) / w);
        A.out = out; A.out_limbs = P.s2; A.batch = B; A.rows = rows;
        DISPATCH_SPLIT(M.G, M.L, (run_multi_lookup<GG, LL>(A)));
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// csrc/primality.h: strong-probable-prime test of n[i] to base[i] (rows of `limbs` 32-bit words), one modulus per row;
// 16-lane groups like the product (kernels_mr.hip).  Returns 2 if no compiled geometry covers 32*limbs + 4 bits.
int emu_miller_rabin(const uint32_t* n, const uint32_t* base, int limbs, uint8_t* pass, uint64_t B) {
    try {
        if (B == 0) return 0;
        const host::Geometry geo = host::pick_geometry(32 * limbs, 0, 16);
        if (geo.G != 16) return 2;
        MillerRabinArgs A;
        A.n = n; A.base = base; A.limbs = limbs; A.pass = pass; A.batch = B;
        DISPATCH_GL(geo.G, geo.L, (run_miller_rabin<GG, LL>(A)));
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// csrc/radix_conv.h on plain arrays: the per-number routines k_to_decimal / k_from_decimal run per thread
struct PlainWords {
    uint32_t* p;
    uint32_t& operator()(int j) const { return p[j]; }
};
int emu_decimal_width(int words) { return decimal_width(words); }
// returns the first row that does not fit `width` digits, or -1
long long emu_to_decimal(const uint32_t* limbs, int words, char* digits, int width, uint64_t B) {
    long long bad = -1;
    std::vector<uint32_t> tmp((size_t)words);
    for (uint64_t i = 0; i < B; ++i) {
        memcpy(tmp.data(), limbs + i * (uint64_t)words, (size_t)words * 4);
        if (!limbs_to_decimal(PlainWords{tmp.data()}, words, digits + i * (uint64_t)width, width) && bad < 0) bad = (long long)i;
    }
    return bad;
}
// returns 0, or 1 (not a digit) / 2 (too large) with *bad_row = the first offending row (the lower row wins)
int emu_from_decimal(const char* digits, int width, uint32_t* limbs, int words, uint64_t B, uint64_t* bad_row) {
    int status = 0;
    for (uint64_t i = 0; i < B; ++i) {
        const int st = decimal_to_limbs(digits + i * (uint64_t)width, width, PlainWords{limbs + i * (uint64_t)words}, words);
        if (st && !status) { status = st; *bad_row = i; }
    }
    return status;
}

// geometry of the split-modulus kernels for modulus n (given as `limbs` words): GL_out = {G, L}; G = 0 if none
// the geometry the CRT halves of this key run on (rung 0 of the private side: one lane per number where p, q are short enough)
int emu_private_split_geometry(const uint32_t* p, const uint32_t* q, const uint32_t* hp, const uint32_t* hq, const uint32_t* p_inverse,
                               int pq_limbs, int n_limbs, int* GL_out) {
    try {
        host::PrivatePlan P = host::build_private(p, q, hp, hq, p_inverse, pq_limbs, n_limbs, g_prefer_group);
        GL_out[0] = P.psplit.G;
        GL_out[1] = P.psplit.L;
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

int emu_split_geometry(const uint32_t* n, int limbs, int* GL_out) {
    try {
        const host::Geometry geo = host::pick_geometry_split(host::big_bits(host::big_from(n, limbs, limbs)), g_prefer_group);
        GL_out[0] = geo.G; GL_out[1] = geo.L;
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// geometry key_setup.h picks for an arbitrary odd modulus given as `limbs` 32-bit words: GL_out = {G, L}
int emu_modulus_geometry(const uint32_t* N, int limbs, int* GL_out) {
    try {
        host::ModulusPack M = host::build_modulus(host::big_from(N, limbs, limbs), nullptr, 32 * limbs, g_prefer_group);
        GL_out[0] = M.G; GL_out[1] = M.L;
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// key-setup introspection: which = 0:n 1:r1 2:r2 3:r3 4:aux (S words of 29-bit limbs each); returns S
int emu_public_constants(const uint32_t* n, int n_limbs, int which, uint32_t* out, int* GL_out, uint32_t* n0inv,
                         int* sched_info /* window, tbl_entries, first_idx, n_ops, squarings, multiplies */) {
    try {
        host::PublicPlan P = host::build_public(n, n_limbs, g_prefer_group);
        const std::vector<uint32_t>* src[] = {&P.nsq.n, &P.nsq.r1, &P.nsq.r2, &P.nsq.r3, &P.nsq.aux};
        memcpy(out, src[which]->data(), sizeof(uint32_t) * (size_t)P.nsq.S);
        GL_out[0] = P.nsq.G; GL_out[1] = P.nsq.L;
        *n0inv = P.nsq.n0inv;
        sched_info[0] = P.exp_n.window; sched_info[1] = P.exp_n.tbl_entries; sched_info[2] = P.exp_n.first_idx;
        sched_info[3] = (int)P.exp_n.ops.size(); sched_info[4] = P.exp_n.squarings; sched_info[5] = P.exp_n.multiplies;
        return P.nsq.S;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

}  // extern "C"
