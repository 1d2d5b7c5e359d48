// mul_table.h — a*b mod N (phe/util.py:53-64 mulmod; phe/paillier.py:705-719 _raw_add) as ONE plain product and ONE fold
// against a per-key table: half the multiply-adds of the two Montgomery products mul_io.h spends on it.
//
// mul_io.h computes a*b*R^-1 (S^2 multiply-adds for the product, S^2 for the reduction) and then repairs the factor R^-1
// with a second Montgomery product by R^2: 4 S^2 for what the reference asks for once.  Plain residues in, plain residue out
// needs no Montgomery form at all:
//     T = a*b = lo + sum_i f_i W^(P+i)            W = 2^29; lo: the P = ceil(bits(N) / 29) low limbs of T, f_i the limbs above
//     y = lo + sum_i f_i * C_i                    C_i = W^(P+i) mod N, a batch-constant row per i (the TABLE, in LDS)
// y = T (mod N) and y < (D + 1) 2^29 N < 2^37 N for the D <= 2^7 fold digits of a 2048-bit key: one quotient estimate (a
// double-precision product on the top limbs), r = y - q N < 3 N, and the usual conditional subtractions give the canonical
// residue — the same bits gmpy2.mod(gmpy2.mul(a, b), c) returns.  S^2 multiply-adds for the product (mul_wide), D*S for the
// fold, 2 S for the quotient: 2 S^2 instead of 4 S^2, and the fold is a chain-free accumulation: no quotient digit, no
// cross-lane step per row, every multiply-add independent of its neighbours.
//
// What it costs: the table is D rows of S limbs — 81 KB for a 2048-bit key's n^2 — and every row is read once per element:
// it lives in LDS (all groups of a wave read the same 16 slices: broadcast reads), which takes a 512-thread workgroup per CU
// with the whole 160 KB to itself (table | constants | one digit row per group | the LDS-DMA staging of mul_io.h).  Keys whose
// table does not fit (from ~2800 bits on) keep mul_io.h's kernels.
//
// Geometry: 16-lane groups only (S = 16 L limbs, L = 5 or 9); written against the same wave:: primitives as mont_core.h.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "mont_core.h"
#include "mul_io.h"
#include "split_core.h"

namespace phe {

struct TableMulArgs {
    const uint32_t* n;       // N, S limbs of 29 bits
    const uint32_t* ncomp;   // W^S - N
    const uint32_t* ncomp1;  // (W^S - N) * W mod W^S
    const uint32_t* table;   // `digits` rows of S words in the device layout (table_row_limbs below)
    double inv;              // W^base / N
    int split;               // P: limbs of T below it are kept, the ones above are folded
    int digits;              // D: fold digits = (S - P) top limbs of the low half + the limbs the high half can have
    int digits_padded;       // mul_tile.h: D rounded up to a multiple of 8 (the column-block table has zero rows from D on)
    int tile_waves;          // mul_tile.h: column blocks the table was cut into (= kTileWaves of the kernel)
    int base;                // limb index the quotient estimate reads y from (4 limbs: base ... base + 3)
    const uint32_t* a;
    const uint32_t* b;
    uint32_t* out;
    size_t a_stride, b_stride, out_stride;  // 32-bit words between consecutive rows (16-byte aligned rows)
    int limbs;                              // 32-bit words per number (a multiple of 4)
    uint64_t batch;
#if defined(PHE_TILE_PROFILE)
    uint64_t* profile;  // measurement-only build: 8 sums of shader clocks (mul_tile.h PHE_TILE_MARK)
#endif
};

constexpr int kTableRowSlack = 16;  // a group's digit row holds S + this many words (the fold digits are S - P more than S)

// words of a 512-thread workgroup's LDS: table | n, ncomp, ncomp1 | 32 digit rows | 8 waves x (stage a | stage b)
template <int L>
constexpr int table_lds_words(int digits) {
    return digits * 16 * L + 3 * 16 * L + 32 * (16 * L + kTableRowSlack) + 8 * 2 * RowIO<16, L>::kStageWave;
}

// Row i of the table as the device wants it: the L limbs of lane g in 16-byte pieces, piece c of all lanes side by side
// ([piece][lane][4 words], the remainder [lane][L % 4 words]) — every LDS read of a lane is an aligned 16-byte read and the 16
// lanes of a group read 256 consecutive bytes.  (key_setup.h:build_table_mul writes this layout.)
template <int L>
PHE_DEV void table_row_limbs(uint32_t (&t)[L], const uint32_t* row, uint32_t g) {
    constexpr int kFull = L / 4, kRem = L % 4;
#pragma unroll
    for (int c = 0; c < kFull; ++c) {
        const Words4 w = *reinterpret_cast<const Words4*>(row + c * 64 + 4 * (int)g);
        t[4 * c] = w.x;
        t[4 * c + 1] = w.y;
        t[4 * c + 2] = w.z;
        t[4 * c + 3] = w.w;
    }
#pragma unroll
    for (int r = 0; r < kRem; ++r) t[4 * kFull + r] = row[kFull * 64 + (int)g * kRem + r];
}

// 64-bit column sums -> columns below 2^29 + what the lane below hands up (value unchanged; the top lane's carry is 0 for
// values < W^S): keeps the chain-free accumulation of the fold below 2^64
template <int L>
PHE_DEV void table_renormalize(uint64_t (&acc)[L], const Lanes<16>& ln) {
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < L; ++k) {
        const uint64_t v = acc[k] + carry;
        acc[k] = v & kLimbMask;
        carry = v >> kRadixBits;
    }
    const uint32_t lo = wave::grp_up1<16>((uint32_t)carry, ln);
    const uint32_t hi = wave::grp_up1<16>((uint32_t)(carry >> 32), ln);
    acc[0] += ((uint64_t)hi << 32) | lo;
}

// row  : the group's digit row (16 L + kTableRowSlack words)
// stage: the WAVE's staging area, 2 * RowIO<16, L>::kStageWave words (a | b); a's half doubles as the low half of the product
// tbl  : the table in LDS;  cst: n | ncomp | ncomp1 in LDS (S words each)
template <int L>
PHE_DEV void mul_table_body(const TableMulArgs& A, uint32_t* row, uint32_t* stage, const uint32_t* tbl, const uint32_t* cst,
                            uint32_t slot, uint32_t total_slots, uint32_t lane) {
    constexpr int G = 16, S = G * L;
    using IO = RowIO<G, L>;
    static_assert(4 * ((S + 3) & ~3) <= IO::kStageWave, "the low halves of a wave's four products must fit the staging area of one operand");
    const Lanes<G> ln(lane);
    const uint32_t g = ln.g, gw = lane / G;
    uint32_t* stage_a = stage;
    uint32_t* stage_b = stage + IO::kStageWave;
    uint32_t* lo_row = stage_a + gw * ((S + 3) & ~3);  // (free once a's digits have left the staging area)
    uint32_t n[L];
    load_row<L>(n, cst, g);
    const int P = A.split, D = A.digits, n_lo = S - P;
    PHE_BOUNDS(n_lo >= 2 && n_lo <= kTableRowSlack && D <= n_lo + S && A.base >= 0 && A.base + 3 < S && (int)gw < 4);
#pragma unroll
    for (int t = 0; t < 2 * IO::kVec; ++t) {  // chunks at or beyond the row length are never copied: they must read as zero
        Words4 z;
        z.x = z.y = z.z = z.w = 0u;
        *reinterpret_cast<Words4*>(stage + t * 256 + 4 * (int)lane) = z;
    }
    wave::lds_fence();
    const uint64_t n_iter = (A.batch + total_slots - 1) / total_slots;
    {
        const uint64_t first = (slot < A.batch) ? slot : A.batch - 1;
        stage_row_async<G, L>(stage_a, A.a + first * A.a_stride, A.limbs, g);
        stage_row_async<G, L>(stage_b, A.b + first * A.b_stride, A.limbs, g);
    }
    for (uint64_t it = 0; it < n_iter; ++it) {
        uint64_t item = slot + it * (uint64_t)total_slots;
        const bool live = item < A.batch;
        if (!live) item = A.batch - 1;
        uint32_t y[L], hi[L], zero[L];
        wave::wait_async_copies();
        {
            const uint32_t gi = wave::reread(g), gwi = wave::reread(gw);
            limbs_from_stage<G, L>(y, stage_b, gwi, gi);
            digits_from_stage<G, L>(row, stage_a, gwi, gi);
        }
        wave::lds_fence();
        // ---- T = a*b: hi * W^S + lo (lo: S canonical digits in LDS, hi: almost-normalised in registers) ---------------------
#pragma unroll
        for (int k = 0; k < L; ++k) zero[k] = 0u;
        mul_wide<G, L>(hi, row, y, zero, lo_row, ln, S);
        wave::lds_fence();
        // ---- the fold digits: the top S - P digits of lo, then the limbs of hi; the kept part of lo opens the accumulators ----
        uint64_t acc[L];
        {
            const uint32_t gi = wave::reread(g);
            if (gi == 0u)
                for (int j = 0; j < n_lo; ++j) row[j] = lo_row[P + j];
#pragma unroll
            for (int k = 0; k < L; ++k) {
                row[n_lo + (int)gi * L + k] = hi[k];
                acc[k] = ((int)gi * L + k < P) ? lo_row[(int)gi * L + k] : 0u;
            }
        }
        wave::lds_fence();
        {   // the low halves were written across a's staging area: the chunks a copy never touches (words at or beyond the row
            // length) must read as zero again before the next row is sliced out of it
            const uint32_t gi = wave::reread(g);
#pragma unroll
            for (int t = 0; t < IO::kVec; ++t) {
                if (4 * (t * G + (int)gi) >= A.limbs) {
                    Words4 z;
                    z.x = z.y = z.z = z.w = 0u;
                    *reinterpret_cast<Words4*>(stage_a + t * 256 + 4 * (int)lane) = z;
                }
            }
        }
        wave::lds_fence();
        if (it + 1 < n_iter) {  // the next element's rows: copied while this one is folded (both staging halves are free now)
            uint64_t nxt = slot + (it + 1) * (uint64_t)total_slots;
            if (nxt >= A.batch) nxt = A.batch - 1;
            const uint32_t gi = wave::reread(g);
            stage_row_async<G, L>(stage_a, A.a + nxt * A.a_stride, A.limbs, gi);
            stage_row_async<G, L>(stage_b, A.b + nxt * A.b_stride, A.limbs, gi);
        }
        // ---- y = lo_kept + sum_i f_i * C_i: chain-free; 48 products of < 2^58.01 per column between two renormalisations ----
        {
            const uint32_t gi = wave::reread(g);
            int i = 0, since = 0;  // digits folded since the accumulators were last brought below 2^29 + carries
#pragma unroll 1
            for (; i + 4 <= D; i += 4) {
                const Words4 dg = *reinterpret_cast<const Words4*>(row + i);
                const uint32_t d4[4] = {dg.x, dg.y, dg.z, dg.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    uint32_t t[L];
                    table_row_limbs<L>(t, tbl + (size_t)(i + u) * S, gi);
#pragma unroll
                    for (int k = 0; k < L; ++k) acc[k] = wave::mad64(d4[u], t[k], acc[k]);
                }
                since += 4;
                if (since == 48) {  // 48 products of < 2^58.01 on top of < 2^37: below 2^64 whatever the operands
                    table_renormalize<L>(acc, ln);
                    since = 0;
                }
            }
            for (; i < D; ++i) {
                uint32_t t[L];
                table_row_limbs<L>(t, tbl + (size_t)i * S, gi);
                const uint32_t d = row[i];
#pragma unroll
                for (int k = 0; k < L; ++k) acc[k] = wave::mad64(d, t[k], acc[k]);
            }
        }
        // ---- y canonical; q^ = floor(y / N) - 1 or - 2 (never above) from four limbs of y ----------------------------------------
        uint32_t t[L];
        normalize_partial<G, L>(t, acc, ln);
        normalize_full<G, L>(t, ln);
        lds_put<L>(row, t, g);
        uint32_t q0, q1;
        {
            const double yd = ((double)row[A.base + 3] * 536870912.0 + (double)row[A.base + 2]) * 288230376151711744.0 +
                              ((double)row[A.base + 1] * 536870912.0 + (double)row[A.base]);
            const double qd = __builtin_floor(yd * A.inv);
            const uint64_t q = qd >= 1.0 ? (uint64_t)qd - 1u : 0u;  // (the estimate may be one too high: never let r go negative)
            q0 = (uint32_t)q & kLimbMask;
            q1 = (uint32_t)(q >> kRadixBits);
        }
        // ---- r = y - q^ N = (y + q^ (W^S - N)) mod W^S  <  3 N ------------------------------------------------------------------
        {
            uint32_t c0[L], c1[L];
            load_row<L>(c0, cst + S, g);
            load_row<L>(c1, cst + 2 * S, g);
#pragma unroll
            for (int k = 0; k < L; ++k) acc[k] = wave::mad64(q1, c1[k], wave::mad64(q0, c0[k], (uint64_t)t[k]));
        }
        normalize_partial<G, L>(t, acc, ln);  // (the carry out of the top lane — q^ itself — is the multiple of W^S dropped)
        canonicalize<G, L>(t, n, ln);
        store_words<G, L>(A.out + item * A.out_stride, A.limbs, t, row, wave::reread(g), live);
    }
}

}  // namespace phe
