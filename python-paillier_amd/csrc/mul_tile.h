// mul_tile.h — a*b mod N (phe/util.py:53-64 mulmod; phe/paillier.py:705-719 _raw_add) as one plain product and one fold
// against the key's table (the arithmetic of mul_table.h) with LANE = ELEMENT: a workgroup of W = 16 waves (1024 threads, four
// per SIMD) takes a tile of 64 products, every lane of every wave works on "its" element, and the waves split the COLUMNS of
// the numbers: S = 16 L columns in W blocks of C = S / W = L (9 for 2048-bit keys, 14 for 3072-bit keys).  Round 5: the same body on
// W = 8 waves (512 threads, S = 8 L = 72 columns: n^2 of a 1024-bit key, two workgroups per CU) — TileShape<L, W> below; where this
// header says "16 waves" read W.
//
// Why: in the limb-group form (16 lanes per number) a row of the product is 9 multiply-adds against 13 other instructions
// (digit broadcast, shift across lanes, carries, masks), and the fold reads every table word from LDS once per lane — 4 bytes
// of LDS per multiply-add, twice what a CU's LDS delivers.  Measured (tools/exp/tile_phases.hip, profiles/r04l): the product
// ran at 40 % multiply-add density, the LDS-fed fold at a quarter of the multiply-add peak.  With one element per lane
//   * nothing crosses lanes: the digits of 64 numbers lie side by side in LDS (buffer[row r][lane e]: every read and write of
//     a wave is 256 consecutive bytes), a step of the product is ONE digit of a (LDS) times a sliding window of C digits of b
//     held in registers — C multiply-adds, two LDS words, no shift (the window "moves" by register renaming in the unrolled
//     code), no mask, no broadcast;
//   * a table word of the fold is the same for all 64 lanes: it comes through the scalar data cache into an SGPR and enters
//     v_mad_u64_u32 as its scalar operand (wave::ScalarRow): no LDS, no VGPR, 4 bytes per 64 multiply-adds.  The table (84 KB
//     at 2048 bits, 190 KB at 3072; [wave][digit][C words]) streams from L2; the requests run a group of fold digits ahead of
//     the multiply-adds.
//
// Phases of a tile (wave w owns the column blocks named; `|` = workgroup barrier):
//   load     wave w: digits [C w, C w + C) of a and b of all 64 elements: 16-byte global loads of the lane's own row (issued
//            one tile ahead, under the settle), re-sliced to 29 bits in registers, to A[digit][e], B[digit][e]              |
//   product  wave w: columns [C w, +C) and [C (w + W), +C) of T = a*b — (W + 1) C steps of C multiply-adds, the same for every
//            wave; 64-bit column sums, carries inside the lane, the block's carry-out to LDS                                |
//            the carry-out of the block below enters the block's two lowest digits; T[row][e] over A and B                  |
//   fold     wave w: columns [C w, +C) of y = T_low + sum_i T[P + i] * C_i (the accumulators open from the wave's own low block
//            of the product: the same columns); carries inside the lane, y to LDS as [element][column], the block's
//            carry-out beside it                                                                                            |
//   settle   round 6: in the fold's own layout (lane = element, wave = column block: tile_settle_blocks below) — the block carries
//            and the four limbs of the quotient estimate through LDS |, r = y - floor(y / N) N as y + q (W^S - N) by blocks, its
//            block carries |, a carry look-ahead over the blocks' generate / propagate bits, the digits [digit][element] |, every
//            wave the 32-bit words [wpw w, wpw (w + 1)) of all 64 rows, 16-byte stores.  (Rounds 4-5: 16 lanes per element, a wave's
//            four elements at once — 1,191 instructions per wave and tile, a quarter of the kernel; PHE_VARIANT_SETTLE_ROWS)            |
// Within the product and the fold a wave lowers its issue priority as it gets through its share (wave::set_priority): the waves
// of a SIMD are served oldest first and would otherwise finish one after the other, the last one alone.
// Same bits as mul_table.h, mul_io.h and gmpy2.mod(gmpy2.mul(a, b), c).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <type_traits>

#include "mont_core.h"
#include "mul_io.h"
#include "mul_table.h"
#include "split_core.h"

// measurement-only build (tools/exp/tile_phases.hip): shader-clock time per phase, summed over the tiles of a wave
#if defined(PHE_TILE_PROFILE)
#define PHE_TILE_MARK(i)                                    \
    {                                                       \
        const uint64_t now_ = __builtin_readcyclecounter(); \
        prof_[i] += now_ - last_;                           \
        last_ = now_;                                       \
    }
#else
#define PHE_TILE_MARK(i) ((void)0)
#endif

namespace phe {

constexpr int kTile = 64;        // elements of a tile = lanes of a wave
constexpr int kTileWaves = 16;   // waves of the workgroup (W); 2 W column blocks of S / W columns make the product
                                 // (key_setup.h kTileWavesHost: the table's column blocks)
constexpr int kFoldPadRows = 4;  // zero rows the column-block table carries past its last digit (the fold's look-ahead)
constexpr int kFoldChunk = 48;   // fold digits between two hand-overs of the accumulators' upper halves (48 * 2^58.01 + 2^32 < 2^64)

// W: the waves of the workgroup = the column blocks of a number = the lanes per element of the settle.  W = 16 (1024 threads, one
// workgroup per CU): S = 16 L columns — 2048-bit keys (L = 9), 3072-bit keys (L = 14).  W = 8 (512 threads, round 5): S = 8 L — n^2 of
// a 1024-bit key needs 72 columns (2048 bits + 38 of headroom), exactly 8 x 9, where 16 x 5 = 80 columns cost (80/72)^2 = 1.23x the
// multiply-adds at 5 instead of 9 multiply-adds per step; its 58 KB of LDS let two workgroups share a CU, so one's barriers
// and settle run under the other's product and fold.
template <int L, int W = kTileWaves>
struct TileShape {
    static_assert(W == 16 || W == 8, "the settle runs on limb groups of W lanes: one DPP row, or half of one");
    static constexpr int kWaves = W;
    static constexpr int S = W * L;     // columns of the fold and of the settle; digit rows of an operand (the ones past its
                                        // last digit hold zeros)
    static constexpr int CW = S / W;    // C: columns of a block = what one wave folds = what one lane of the settle holds (L)
    static constexpr int kFoldGroup = CW > 10 ? 2 : 4;  // fold digits per request group (28 ... 40 table words in flight per group)
    static constexpr int kRowT = S + kLdsPad;  // a settle group's digit row
    // rows of the tile buffer: A (S rows + one zero row) | B (S rows); T (2S rows) over both; during the settle the settled
    // columns y[element][column] lie in rows [0, P) and the 64 digit rows of the limb groups from row kSettleRow0 >= P on (the
    // fold's digits there are dead by then).  W = 16: S - 2 (P <= S - 2 there; 3072-bit keys need the rows: 147 KB of LDS);
    // W = 8: S (P may be S - 1: the 71 digits of a 2048-bit n^2 in 72 columns)
    static constexpr int kSettleRow0 = W == 16 ? S - 2 : S;
    static constexpr int kRows = (2 * S + 1 > kSettleRow0 + kRowT) ? 2 * S + 1 : kSettleRow0 + kRowT;
    // 32-bit words wave w must see to cut its digits [CW w, CW w + CW): from the 16-byte piece that holds bit 29 CW w on
    // (rows are whole 16-byte pieces: a piece is either inside the row or beyond it)
    static constexpr int first_word(int w) { return ((kRadixBits * CW * w) >> 5) & ~3; }
    static constexpr int word_skip(int w) { return ((kRadixBits * CW * w) >> 5) & 3; }
    static constexpr int max_chunks() {
        int m = 0;
        for (int w = 0; w < W; ++w) {
            const int c = (word_skip(w) + ((kRadixBits * (CW - 1)) >> 5) + 3 + 3) / 4;
            m = c > m ? c : m;
        }
        return m;
    }
    static constexpr int kChunks = max_chunks();
    static_assert((kTile / W) * W == kTile, "one digit row per limb group of the settle: 64 rows of kRowT words = kRowT rows of the buffer");
    // LDS words: tile buffer | product carries (2 words x 2W blocks x 64) | top columns | fold carries | n, ncomp, ncomp1
    static constexpr int kLdsWords = kRows * kTile + 2 * 2 * W * kTile + kTile * kTableRowSlack + 2 * W * kTile + 3 * S;
};
template <int L, int W = kTileWaves>
constexpr int tile_lds_words() {
    return TileShape<L, W>::kLdsWords;
}

// A column sum of up to 2S products of < 2^58.01 (almost 2^66) is kept as  acc + upper * 2^B:  acc a 64-bit accumulator that is cut
// back below 2^B at every hand-over (so that 48 more products fit), upper the sum of what was cut.  B = 32 costs nothing to cut
// (the accumulator's upper register) but the sum of the cuts needs 34 bits — two registers per column; B = 36 fits 32 bits (one
// shift more per hand-over).  Narrow blocks (2048-bit keys: 9 columns) have the registers and take B = 32 (same box: 2 % faster);
// at 14 columns (3072-bit keys) the second register per column spills inside the product loop and B = 36 is 18 % faster.
template <int CW>
struct TileUpper {
    static constexpr int kBits = CW > 10 ? 36 : 32;
    typedef typename std::conditional<(CW > 10), uint32_t, uint64_t>::type word;
};
template <int CW>
PHE_DEV void tile_hand_over(uint64_t (&acc)[CW], typename TileUpper<CW>::word (&upper)[CW]) {
    constexpr int B = TileUpper<CW>::kBits;
#pragma unroll
    for (int k = 0; k < CW; ++k) {
        upper[k] += (typename TileUpper<CW>::word)(acc[k] >> B);
        acc[k] &= (1ull << B) - 1u;
    }
}
// the column sums of one lane -> digits below 2^29 and the carry that leaves the block (< 2^38)
template <int CW>
PHE_DEV uint64_t tile_block_carries(uint32_t (&digit)[CW], const uint64_t (&acc)[CW], const typename TileUpper<CW>::word (&upper)[CW]) {
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < CW; ++k) {
        const uint64_t v = (acc[k] & 0xffffffffull) + carry;
        digit[k] = (uint32_t)v & kLimbMask;
        carry = (v >> kRadixBits) + ((acc[k] >> 32) << (32 - kRadixBits)) + ((uint64_t)upper[k] << (TileUpper<CW>::kBits - kRadixBits));
    }
    return carry;
}

// blocks of CW products a column accumulator takes between two hand-overs: after one it is below 2^B (B = 32 or 36), K blocks and the
// closing block of a column's low half add (K + 1) CW products of < 2^58 (1 + 2^-20): below 2^64 while (K + 1) CW <= 63
template <int CW>
constexpr int tile_hand_over_blocks() { return (63 - CW) / CW; }
// A wave lowers its issue priority as it gets through its W + 1 blocks of the product (set_priority).  `done` counts them: the level
// 3 - 4 done / (W + 1) changes at three values of `done` — three compares on the scalar unit instead of a division and a ladder of
// branches per block (the scalar unit of a CU serves sixteen waves: round 6 found the fold waiting for it, see the fold below)
template <int W>
PHE_DEV void tile_priority_after(uint32_t done) {
    constexpr uint32_t b1 = (W + 1 + 3) / 4, b2 = (2 * (W + 1) + 3) / 4, b3 = (3 * (W + 1) + 3) / 4;
    if (done == b1) wave::set_priority(2);
    else if (done == b2) wave::set_priority(1);
    else if (done == b3) wave::set_priority(0);
}

// CW steps of a column block of the product.  Before: win[j] = b[q0 + j] (q0 = the block's lowest column minus the step
// index i; digits outside b read as zero), a_col = &A[i][e], b_col = &B[q0 - 1][e].  After: the same for i + CW (q0 - CW).
// Step u multiplies a[i + u] into every column: column c takes b[q0 + c - u], which is win[(c - u) mod CW] once the digits
// b[q0 - 1 ... q0 - u] have replaced win[CW - 1 ... CW - u] — the window slides by renaming, not by moving.  LOADS = false:
// the digits that would enter lie below b[0] (the last CW steps of a low block).
template <int CW, bool LOADS>
PHE_DEV void tile_product_steps(uint64_t (&acc)[CW], const uint32_t (&win)[CW], uint32_t (&nxt)[CW], const uint32_t* a_col, const uint32_t* b_col) {
    // win: the window on entry; nxt: the window the NEXT block starts from — the digits this block lets in land there directly
    // (slot CW - 1 - u at step u), and a step reads a slot from nxt once it was replaced.  The caller alternates the two arrays from block
    // to block, so the window never moves (round 6: CW copies per block of CW^2 multiply-adds gone).
    // the digits of (half) the block first, then the multiply-adds with nothing to wait for (left to place the reads itself the
    // compiler puts each one in front of its first use, with the wait for it).  Wide blocks take the digits in two halves: 2 CW
    // registers for them are what spills at CW = 14.
    constexpr int kParts = CW > 10 ? 2 : 1, kPer = (CW + kParts - 1) / kParts;
    const wave::lds_u32* b_low = wave::reread_lds(b_col - (CW - 1) * kTile);  // (one address, CW immediates: left alone the compiler re-bases every read)
#pragma unroll
    for (int part = 0; part < kParts; ++part) {
        uint32_t ad[kPer];
#pragma unroll
        for (int v = 0; v < kPer; ++v) {
            const int u = part * kPer + v;
            if (u < CW) {
                ad[v] = a_col[u * kTile];
                nxt[CW - 1 - u] = LOADS ? b_low[(CW - 1 - u) * kTile] : 0u;  // (offsets from the block's lowest row: immediates of the LDS reads)
            }
        }
        wave::order_fence();
#pragma unroll
        for (int v = 0; v < kPer; ++v) {
            const int u = part * kPer + v;
            if (u < CW) {
#pragma unroll
                for (int c = 0; c < CW; ++c) {
                    const int slot = (c - u + CW) % CW;  // replaced by the steps before this one iff slot >= CW - u
                    acc[c] = wave::mad64(ad[v], slot >= CW - u ? nxt[slot] : win[slot], acc[c]);
                }
            }
        }
        wave::order_fence();
    }
}

// digits [CW wv, CW wv + CW) of the number whose 32-bit words from TileShape::first_word(wv) on are in `w` (4 per piece); SKIP =
// TileShape::word_skip(wv) words lie before the one that holds the first digit's lowest bit.  One funnel shift by the
// wave-uniform bit offset of that digit, then every digit sits at a compile-time position.
template <int L, int W, int SKIP>
PHE_DEV void tile_cut_digits(uint32_t* column, const Words4 (&w)[TileShape<L, W>::kChunks], uint32_t wv) {
    using T = TileShape<L, W>;
    constexpr int kWords = 4 * T::kChunks;
    uint32_t v[kWords + 1];
#pragma unroll
    for (int c = 0; c < T::kChunks; ++c) {
        v[4 * c] = w[c].x;
        v[4 * c + 1] = w[c].y;
        v[4 * c + 2] = w[c].z;
        v[4 * c + 3] = w[c].w;
    }
    v[kWords] = 0u;
    const uint32_t shift = (kRadixBits * T::CW * wv) & 31u;
    uint32_t* out = column + (size_t)(T::CW * wv) * kTile;
#pragma unroll
    for (int j = 0; j < T::CW; ++j) {
        constexpr int kLast = SKIP + ((kRadixBits * (T::CW - 1)) >> 5) + 2;
        static_assert(kLast <= kWords, "a wave's digits lie in the pieces it loads");
        const int bit = kRadixBits * j, q = SKIP + (bit >> 5), o = bit & 31;
        const uint32_t lo = (uint32_t)((((uint64_t)v[q + 1] << 32) | v[q]) >> shift);
        const uint32_t hi = (uint32_t)((((uint64_t)v[q + 2] << 32) | v[q + 1]) >> shift);
        out[j * kTile] = (uint32_t)((((uint64_t)hi << 32) | lo) >> o) & kLimbMask;
    }
}

// a lane's row of the batch from word w0 (a multiple of 4) on, as the 16-byte pieces its wave cuts its digits from (pieces at
// or beyond the row read as zero)
template <int L, int W>
using TileRaw = Words4[TileShape<L, W>::kChunks];  // (an alias: `Words4 (&raw)[TileShape<L, W>::kChunks]` does not parse as a parameter)
template <int L, int W>
PHE_DEV void tile_request_row(TileRaw<L, W>& raw, const uint32_t* p, int w0, int limbs) {
#pragma unroll
    for (int c = 0; c < TileShape<L, W>::kChunks; ++c) {
        Words4 z;
        z.x = z.y = z.z = z.w = 0u;
        raw[c] = z;
        if (w0 + 4 * c < limbs) raw[c] = *reinterpret_cast<const Words4*>(p + 4 * c);
    }
}

// 32-bit words [wpw wv, wpw (wv + 1)) of every element's row from the digits [digit][element] in the tile buffer, four at a time
template <int L, int W>
PHE_DEV void tile_store_words(const TableMulArgs& A, const uint32_t* tile, uint32_t wv, uint32_t e, uint64_t item, bool live) {
    using T = TileShape<L, W>;
    constexpr int S = T::S;
    constexpr int kMaxWpw = (((kRadixBits * S + 31) / 32 + W - 1) / W + 3) & ~3;  // (S digits never make more words than this per wave)
    const int wpw = ((A.limbs + W - 1) / W + 3) & ~3;
    uint32_t* out = A.out + item * A.out_stride;
#pragma unroll
    for (int gq = 0; gq < kMaxWpw / 4; ++gq) {
        const int j = (int)wv * wpw + 4 * gq;  // (wave-uniform) first word of the group
        if (4 * gq < wpw && j < A.limbs) {
            const int bit = 32 * j, q = bit / kRadixBits, o = bit - q * kRadixBits;
            uint32_t dg[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) dg[i] = (q + i < S) ? tile[(size_t)(q + i) * kTile + e] : 0u;
            // the 174 bits of six digits as 32-bit words at compile-time positions, then one shift by the uniform offset o < 29
            const uint32_t z0 = dg[0] | (dg[1] << 29), z1 = (dg[1] >> 3) | (dg[2] << 26), z2 = (dg[2] >> 6) | (dg[3] << 23),
                           z3 = (dg[3] >> 9) | (dg[4] << 20), z4 = (dg[4] >> 12) | (dg[5] << 17);
            Words4 w4;
            w4.x = (uint32_t)((((uint64_t)z1 << 32) | z0) >> o);
            w4.y = (uint32_t)((((uint64_t)z2 << 32) | z1) >> o);
            w4.z = (uint32_t)((((uint64_t)z3 << 32) | z2) >> o);
            w4.w = (uint32_t)((((uint64_t)z4 << 32) | z3) >> o);
            if (live) *reinterpret_cast<Words4*>(out + j) = w4;
        }
    }
}

// ---- the settle in the fold's own layout: lane = element, wave = column block (round 6) --------------------------------------
// The settle above the fold (16 lanes per element, four elements of a wave at a time) issues 1,191 instructions per wave and tile —
// a quarter of the kernel's — for 64 elements a workgroup; in the fold's layout every instruction works on 64 elements.  What crosses
// the column blocks goes through LDS in four short exchanges:
//   1  the fold's block carries (< 2^38) and the four limbs the quotient estimate reads                                           |
//      every wave: q^ = floor(y / N) - 1 or - 2 (the same double-precision estimate), then THREE candidates r_m = y + (q^ + m) K,
//      m = 0, 1, 2, K = W^S - N (the conditional subtractions of the row settle as additions of the complement: no borrow, no
//      comparison); per candidate the block's digits and its carry-out (< 2^31)
//   2  the candidates' block carries                                                                                              |
//      the carry of the block below enters and ripples through the block; what is left to cross a block boundary is one bit:
//      generate / propagate flags of every block OR-ed into one word per element and candidate; the top block adds whether its
//      carry-out is the candidate's q0 + q1 + m — exactly then y - (q^ + m) N is not negative
//   3  the flag words                                                                                                             |
//      every wave: the carry look-ahead over the blocks (one add and one xor per candidate), the largest valid m, its digits
//      (the second-order carry-in ripples through the block only if some element of the tile has one), digits to LDS
//   4  the digits, [digit][element]                                                                                               |
//      wave w: 32-bit words [wpw w, wpw (w + 1)) of every element's row: one funnel shift by the wave-uniform bit offset, 16-byte stores
// 3 of 5 barriers are new; ~430 instructions per wave and tile.
#if defined(PHE_VARIANT_SETTLE_ROWS)  // measurement only: the round-4 settle on 16 lanes per element
constexpr bool kSettleByBlocks = false;
#else
constexpr bool kSettleByBlocks = true;
#endif

// y: the block's digits after the fold (< 2^29), ycarry: what left the block (< 2^38).  LDS: ycar = fold_carry (2 words per block
// and element), ccar = prod_carry (dead since the product: one word per candidate, block and element), top: 4 x 64 estimate limbs |
// 3 x 64 flag words | 64 words of the top block's verdicts; the tile buffer takes the final digits.  after_first: called once past
// the first barrier (the next tile's row requests).
template <int L, int W, class F>
PHE_DEV void tile_settle_blocks(const TableMulArgs& A, uint32_t* tile, uint32_t* ccar, uint32_t* top, uint32_t* ycar, const uint32_t* cst,
                                uint32_t wv, uint32_t e, uint32_t (&y)[TileShape<L, W>::CW], uint64_t ycarry, uint64_t item, bool live,
                                F&& after_first) {
    using T = TileShape<L, W>;
    constexpr int S = T::S, CW = T::CW;
    static_assert(W <= 16 && CW >= 5, "flag words: 16 generate + 16 propagate bits; at most one block boundary among four estimate limbs");
    uint32_t* const est = top;                 // [4][64]
    uint32_t* const flags = top + 4 * kTile;   // [3][64]
    uint32_t* const verdict = top + 7 * kTile; // [3][64]
    const int c0 = (int)wv * CW;
    // ---- 1: block carries of y, estimate limbs ---------------------------------------------------------------------------------
    ycar[(wv * kTile + e) * 2u] = (uint32_t)ycarry;
    ycar[(wv * kTile + e) * 2u + 1u] = (uint32_t)(ycarry >> 32);
    {
        const int j0 = A.base - c0;  // (wave-uniform) this block's digit k is estimate limb k - j0
#pragma unroll
        for (int k = 0; k < CW; ++k)
            if ((unsigned)(k - j0) < 4u) est[(k - j0) * kTile + (int)e] = y[k];
    }
    if (wv == 0u) {
        flags[e] = 0u;
        flags[kTile + e] = 0u;
        flags[2 * kTile + e] = 0u;
    }
    wave::block_barrier();
    after_first();
    // the carry of the block below enters the two lowest digits (digit 1 stays below 2^29 + 2^10: it is only added from here on)
    if (wv != 0u) {
        const uint64_t c = ((uint64_t)ycar[((wv - 1u) * kTile + e) * 2u + 1u] << 32) | ycar[((wv - 1u) * kTile + e) * 2u];
        const uint32_t d0 = y[0] + ((uint32_t)c & kLimbMask);
        y[0] = d0 & kLimbMask;
        y[1] += (uint32_t)(c >> kRadixBits) + (d0 >> kRadixBits);
    }
    // ---- the quotient estimate: four limbs of y as they left their blocks, plus the one block carry that enters among them ---------
    uint32_t q0, q1;
    bool fast;
    {
        double yd = ((double)est[3 * kTile + (int)e] * 536870912.0 + (double)est[2 * kTile + (int)e]) * 288230376151711744.0 +
                    ((double)est[kTile + (int)e] * 536870912.0 + (double)est[e]);
        const int wb = (A.base + CW - 1) / CW, off = wb * CW - A.base;  // first block that starts at or above limb `base`
        if (wb >= 1 && wb < W && off <= 3) {                            // (wave-uniform) its carry-in has the weight W^off
            const uint64_t c = ((uint64_t)ycar[((uint32_t)(wb - 1) * kTile + e) * 2u + 1u] << 32) | ycar[((uint32_t)(wb - 1) * kTile + e) * 2u];
            const double w = off == 0 ? 1.0 : (off == 1 ? 536870912.0 : (off == 2 ? 288230376151711744.0 : 154742504910672534362390528.0));
            yd += (double)c * w;
        }
        const double qe = yd * A.inv, qd = __builtin_floor(qe), frac = qe - qd;
        // The estimate is within 2^-13 of y / N (y / N < 2^38; the limbs below `base` weigh < 2^-70 of a unit; W^base / N, the sum of the
        // four limbs and the product are rounded to 53 bits: 3 x 2^-15).  Where its fraction keeps 2^-11 away from 0 and 1 the floor IS
        // floor(y / N): ONE candidate, r = y - floor N, already below N.  Every wave sees the same limbs of all 64 elements, so the
        // workgroup agrees without asking; a tile with an element too close to call (1 in 2^10) takes the three candidates.
        fast = wave::ballot(!(qd >= 1.0 && frac > 0.00048828125 && frac < 0.99951171875)) == 0;
        const uint64_t q = fast ? (uint64_t)qd : (qd >= 1.0 ? (uint64_t)qd - 1u : 0u);  // (the estimate may be one too high: never let r go negative)
        q0 = (uint32_t)q & kLimbMask;
        q1 = (uint32_t)(q >> kRadixBits);
    }
    if (fast) {
        // ---- one candidate r = y + q K (mod W^S): three exchanges instead of four ---------------------------------------------------
        uint32_t f[CW];
        {
            uint64_t carry = 0;
#pragma unroll
            for (int k = 0; k < CW; ++k) {
                const uint64_t v = wave::mad64(q1, cst[2 * S + c0 + k], wave::mad64(q0, cst[S + c0 + k], (uint64_t)y[k])) + carry;
                f[k] = (uint32_t)v & kLimbMask;
                carry = v >> kRadixBits;
            }
            ccar[wv * kTile + e] = (uint32_t)carry;  // < 2^31
        }
        wave::block_barrier();
        {
            uint32_t c = wv != 0u ? ccar[(wv - 1u) * kTile + e] : 0u;
            uint32_t ones = kLimbMask;
#pragma unroll
            for (int k = 0; k < CW; ++k) {
                const uint32_t v = f[k] + c;
                f[k] = v & kLimbMask;
                c = v >> kRadixBits;
                ones &= f[k];
            }
            if (wv != (uint32_t)W - 1u) {  // (what leaves the top block is the multiple of W^S dropped: q0 + q1)
                const uint32_t word = (c << wv) | ((ones == kLimbMask ? 1u : 0u) << (16u + wv));
                if (wave::ballot(word != 0u) != 0) wave::lds_or(flags + (int)e, word);
            } else {
                PHE_BOUNDS(ccar[wv * kTile + e] + c == q0 + q1 || (ones == kLimbMask && ccar[wv * kTile + e] + c + 1u == q0 + q1));
            }
        }
        uint32_t* rows = tile + (size_t)c0 * kTile + e;
#pragma unroll
        for (int k = 0; k < CW; ++k) rows[k * kTile] = f[k];
        wave::block_barrier();
        {
            const uint32_t w = flags[e];
            if (wave::ballot(w != 0u) != 0) {  // (rare, and the same in every wave: some block carried out once more, or is all ones)
                const uint32_t gen = w & 0xffffu, prop = w >> 16;
                uint32_t c = ((((gen << 1) + prop) ^ prop) >> wv) & 1u;
#pragma unroll
                for (int k = 0; k < CW; ++k) {
                    const uint32_t v = f[k] + c;
                    f[k] = v & kLimbMask;
                    c = v >> kRadixBits;
                    rows[k * kTile] = f[k];
                }
                wave::block_barrier();
            }
        }
        tile_store_words<L, W>(A, tile, wv, e, item, live);
        return;
    }
    // ---- three candidates r_m = y + (q^ + m) K  (mod W^S), K = W^S - N, m = 2, 1, 0: one after the other (a tile in sixteen comes here:
    // registers and code count, barriers do not); every element keeps the digits of the first one that is valid ------------------------
    uint32_t f[CW];
#pragma unroll
    for (int k = 0; k < CW; ++k) f[k] = 0u;
    uint32_t found = 0u;  // all ones once a candidate was valid
#pragma unroll 1
    for (int m = 2; m >= 0; --m) {
        uint32_t d[CW];
        uint32_t cout;
        {
            const uint32_t q0m = q0 + (uint32_t)m;  // (q^ + m) K = (q0 + m) K + q1 (K W): still a 32-bit multiplier
            uint64_t carry = 0;
#pragma unroll
            for (int k = 0; k < CW; ++k) {
                const uint64_t v = wave::mad64(q1, cst[2 * S + c0 + k], wave::mad64(q0m, cst[S + c0 + k], (uint64_t)y[k])) + carry;
                d[k] = (uint32_t)v & kLimbMask;
                carry = v >> kRadixBits;
            }
            cout = (uint32_t)carry;  // < 2^31
            ccar[((uint32_t)m * W + wv) * kTile + e] = cout;
        }
        wave::block_barrier();
        {   // the carry of the block below ripples through the block; one bit is left to cross a boundary
            uint32_t c = wv != 0u ? ccar[((uint32_t)m * W + wv - 1u) * kTile + e] : 0u;
            uint32_t ones = kLimbMask;
#pragma unroll
            for (int k = 0; k < CW; ++k) {
                const uint32_t v = d[k] + c;  // < 2^29 + 2^31
                d[k] = v & kLimbMask;
                c = v >> kRadixBits;
                ones &= d[k];
            }
            if (wv != (uint32_t)W - 1u) {
                const uint32_t word = (c << wv) | ((ones == kLimbMask ? 1u : 0u) << (16u + wv));  // (disjoint: a block that carried out is small)
                if (wave::ballot(word != 0u) != 0) wave::lds_or(flags + m * kTile + (int)e, word);
            } else {
                // what leaves the top block when y - (q^ + m) N is not negative: K W loses its top digit W - 1 in ncomp1, so
                // (q^ + m) W^S - q1 (W - 1) W^S = (q0 + q1 + m) W^S
                const uint32_t expect = q0 + q1 + (uint32_t)m, ctop = cout + c;
                // bit 0: valid if no carry enters the top block; bit 1: if one does (it leaves again only through digits all ones)
                verdict[m * kTile + (int)e] = (ctop == expect ? 1u : 0u) | ((ctop + (ones == kLimbMask ? 1u : 0u) == expect ? 1u : 0u) << 1);
            }
        }
        wave::block_barrier();
        {   // look-ahead over the blocks; the candidate's digits if it is the first valid one
            const uint32_t w = flags[m * kTile + (int)e];
            const uint32_t gen = w & 0xffffu, prop = w >> 16;
            const uint32_t into = (((gen << 1) + prop) ^ prop);  // bit b: a carry enters block b
            const uint32_t valid = 0u - ((verdict[m * kTile + (int)e] >> ((into >> (W - 1)) & 1u)) & 1u);
            uint32_t c = (into >> wv) & 1u;
            if (wave::ballot(c != 0u) != 0) {  // (rare: a block of all ones above a block that carried out)
#pragma unroll
                for (int k = 0; k < CW; ++k) {
                    const uint32_t v = d[k] + c;
                    d[k] = v & kLimbMask;
                    c = v >> kRadixBits;
                }
            }
            const uint32_t take = valid & ~found;
#pragma unroll
            for (int k = 0; k < CW; ++k) f[k] = (d[k] & take) | (f[k] & ~take);
            found |= valid;
        }
    }
    PHE_BOUNDS(found == 0xffffffffu);  // (m = 0 is valid whenever the estimate's error bound holds)
    {
        uint32_t* rows = tile + (size_t)c0 * kTile + e;
#pragma unroll
        for (int k = 0; k < CW; ++k) rows[k * kTile] = f[k];
    }
    wave::block_barrier();
    // ---- 4: 32-bit words [wpw wv, wpw (wv + 1)) of every element's row ------------------------------------------------------------------
    tile_store_words<L, W>(A, tile, wv, e, item, live);
}

// A.table: the fold table in the COLUMN-BLOCK layout [wave w][digit i][C words]: limbs [C w, C (w + 1)) of W^(P+i) mod N,
// A.digits_padded + kFoldPadRows rows per wave (key_setup.h:build_table_mul writes both layouts).
// tile: TileShape::kRows * 64 words; prod_carry: 2 * 2W * 64; top: 64 * kTableRowSlack; fold_carry: 2 * W * 64; cst: n | ncomp |
// ncomp1 (S limbs each).  `wv` must be wave-uniform; every wave of the workgroup runs the
// same number of tiles (the barriers).  Rows of a, b, out: A.limbs words (a multiple of 4), 16-byte aligned.
template <int L, int W = kTileWaves>
PHE_DEV void mul_tile_body(const TableMulArgs& A, uint32_t* tile, uint32_t* prod_carry, uint32_t* top, uint32_t* fold_carry,
                           const uint32_t* cst, uint32_t wv, uint32_t block, uint32_t n_blocks, uint32_t lane) {
    using T = TileShape<L, W>;
    constexpr int S = T::S, CW = T::CW, GS = W;  // (GS: lanes per element of the settle — lane g holds columns [L g, L g + L))
    constexpr int kTileWaves = W;                // (shadows the 16-wave constant: everything below is written in the waves of THIS shape)
    const int P = A.split, D = A.digits_padded;
    PHE_BOUNDS(S - P >= 1 && S - P <= kTableRowSlack && P + D + kFoldPadRows + 2 <= T::kRows + kTableRowSlack && P <= T::kSettleRow0 && A.base >= 0 &&
               A.base + 3 < S && wv < (uint32_t)kTileWaves && A.limbs % 4 == 0 && 32 * A.limbs <= kRadixBits * S);
    uint32_t* const buf_a = tile;                    // A[digit][e]: rows 0 .. S - 1, row S zero
    uint32_t* const buf_b = tile + (S + 1) * kTile;  // B[digit][e]: rows 0 .. S - 1 (B[-1] is A's zero row)
    const uint64_t n_tiles = (A.batch + kTile - 1) / kTile;
    const int w0 = ((kRadixBits * CW * (int)wv) >> 5) & ~3;  // (wave-uniform) the 16-byte piece with the lowest bit of this wave's first digit
    Words4 raw_a[T::kChunks], raw_b[T::kChunks];
    auto request_rows = [&](uint64_t tile_i) __attribute__((always_inline)) {
        uint64_t item = tile_i * kTile + wave::reread(lane);
        if (item >= A.batch) item = A.batch - 1;
        tile_request_row<L, W>(raw_a, A.a + item * A.a_stride + w0, w0, A.limbs);
        tile_request_row<L, W>(raw_b, A.b + item * A.b_stride + w0, w0, A.limbs);
    };
    if (block < n_tiles) request_rows(block);
#if defined(PHE_TILE_PROFILE)
    uint64_t prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_readcyclecounter();
#endif
    for (uint64_t tile_i = block; tile_i < n_tiles; tile_i += n_blocks) {
        const uint32_t e = wave::reread(lane);
        // ---- load: this wave's digits of a and b, all 64 elements ------------------------------------------------------------------
        switch (((kRadixBits * CW * wv) >> 5) & 3u) {  // (wave-uniform: words of the first piece before the first digit)
#define PHE_TILE_CUT(SKIP)                                \
    case SKIP:                                            \
        tile_cut_digits<L, W, SKIP>(buf_a + e, raw_a, wv);   \
        tile_cut_digits<L, W, SKIP>(buf_b + e, raw_b, wv);   \
        break;
            PHE_TILE_CUT(0) PHE_TILE_CUT(1) PHE_TILE_CUT(2) PHE_TILE_CUT(3)
#undef PHE_TILE_CUT
            default: break;
        }
        if (wv == 0u) buf_a[S * kTile + e] = 0u;
        wave::block_barrier();
        PHE_TILE_MARK(0);  // load
        wave::set_priority(3);
        // ---- product: column blocks wv (low) and wv + W (high) of T = a*b ---------------------------------------------------------
        uint32_t t_low[CW], t_high[CW];
        uint64_t out_low, out_high;
        constexpr bool kParkLow = CW > 10;
        [[maybe_unused]] volatile uint32_t parked[kParkLow ? CW : 1];
        {
            uint64_t acc[CW];
            typename TileUpper<CW>::word upper[CW];
            uint32_t win[CW];
            // low block: columns p0 = CW wv ...; steps i = 0 .. p0 + CW - 1, the last CW of them with nothing left of b to enter
            {
                const wave::lds_u32* b_open = wave::reread_lds(buf_b + (size_t)(wv * CW) * kTile + e);  // (one address, CW immediates)
#pragma unroll
                for (int c = 0; c < CW; ++c) {
                    acc[c] = 0;
                    upper[c] = 0;
                    win[c] = b_open[c * kTile];
                }
            }
            {
                const uint32_t* a_col = buf_a + e;
                const uint32_t* b_col = buf_b + ((int)wv * CW - 1) * kTile + e;
                // `blocks` blocks with digits of b entering, two per trip: the window alternates between win and alt and is in win
                // again at the loop's back edge (an odd block at the end copies it back: once per column half, not once per block)
                int since = 0;
                const auto block = [&](const uint32_t (&from)[CW], uint32_t (&to)[CW], uint32_t done) __attribute__((always_inline)) {
                    tile_priority_after<W>(done);  // (blocks done of W + 1)
                    tile_product_steps<CW, true>(acc, from, to, a_col, b_col);
                    a_col += CW * kTile;
                    b_col -= CW * kTile;
                    if (++since == tile_hand_over_blocks<CW>()) {
                        tile_hand_over<CW>(acc, upper);
                        since = 0;
                    }
                };
                uint32_t alt[CW];
                uint32_t t = 0;
#pragma unroll 1
                for (; t + 2u <= wv; t += 2u) {
                    block(win, alt, t);
                    block(alt, win, t + 1u);
                }
                if (t < wv) {
                    block(win, alt, t);
#pragma unroll
                    for (int c = 0; c < CW; ++c) win[c] = alt[c];
                }
                tile_priority_after<W>(wv);
                tile_product_steps<CW, false>(acc, win, alt, a_col, b_col);
            }
            out_low = tile_block_carries<CW>(t_low, acc, upper);
            if constexpr (kParkLow) {
                // wide blocks (3072-bit keys: 14 columns): the low block's digits leave the registers while the high block is multiplied —
                // ONE store and one load per digit and tile in the lane's private memory, where the compiler, short of these CW
                // registers, spilled a dozen values in every block of the high half's loop (138 scratch instructions per tile)
#pragma unroll
                for (int c = 0; c < CW; ++c) parked[c] = t_low[c];
            }
            // high block: columns p0 = CW (wv + W) ...; steps i = p0 - (S - 1) ... : the window opens on b[S - 1] alone
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                acc[c] = 0;
                upper[c] = 0;
                win[c] = 0u;
            }
            win[0] = buf_b[(S - 1) * kTile + (int)e];
            {
                const int i0 = CW * (int)wv + 1;  // (the last of the CW (W - wv) steps reads a[S]: the zero row)
                const uint32_t* a_col = buf_a + i0 * kTile + e;
                const uint32_t* b_col = buf_b + (S - 2) * kTile + e;
                int since = 0;
                const auto block = [&](const uint32_t (&from)[CW], uint32_t (&to)[CW], uint32_t done) __attribute__((always_inline)) {
                    tile_priority_after<W>(done);
                    tile_product_steps<CW, true>(acc, from, to, a_col, b_col);
                    a_col += CW * kTile;
                    b_col -= CW * kTile;
                    if (++since == tile_hand_over_blocks<CW>()) {
                        tile_hand_over<CW>(acc, upper);
                        since = 0;
                    }
                };
                uint32_t alt[CW];
                const uint32_t n_high = (uint32_t)kTileWaves - wv;
                uint32_t t = 0;
#pragma unroll 1
                for (; t + 2u <= n_high; t += 2u) {
                    block(win, alt, wv + 1u + t);
                    block(alt, win, wv + 2u + t);
                }
                if (t < n_high) block(win, alt, wv + 1u + t);  // (the window is not looked at again)
            }
            out_high = tile_block_carries<CW>(t_high, acc, upper);
        }
        prod_carry[(wv * kTile + e) * 2u] = (uint32_t)out_low;
        prod_carry[(wv * kTile + e) * 2u + 1u] = (uint32_t)(out_low >> 32);
        prod_carry[((wv + kTileWaves) * kTile + e) * 2u] = (uint32_t)out_high;
        prod_carry[((wv + kTileWaves) * kTile + e) * 2u + 1u] = (uint32_t)(out_high >> 32);
        PHE_TILE_MARK(1);  // product
        wave::block_barrier();  // every wave is through with A and B, every carry-out is in LDS
        PHE_TILE_MARK(2);  // barrier
        if constexpr (kParkLow) {
#pragma unroll
            for (int c = 0; c < CW; ++c) t_low[c] = parked[c];
        }
        {
            // what left the block below enters the two lowest digits (digit 1 stays below 2^29 + 2^10: almost-normalised, as
            // the fold's bound wants it); T over A and B
            auto enter = [&](uint32_t (&t)[CW], uint32_t blk) {
                const uint64_t c = ((uint64_t)prod_carry[((blk - 1u) * kTile + e) * 2u + 1u] << 32) | prod_carry[((blk - 1u) * kTile + e) * 2u];
                const uint32_t d0 = t[0] + ((uint32_t)c & kLimbMask);
                t[0] = d0 & kLimbMask;
                t[1] += (uint32_t)(c >> kRadixBits) + (d0 >> kRadixBits);
            };
            if (wv != 0u) enter(t_low, wv);
            enter(t_high, wv + kTileWaves);
            // (the low block's columns below P never reach LDS: they open this wave's own fold accumulators — same columns)
            wave::lds_u32* t_rows = wave::reread_lds(tile + (size_t)(wv * CW) * kTile + e);
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                if ((int)wv * CW + c >= P) t_rows[c * kTile] = t_low[c];
                t_rows[(kTileWaves * CW + c) * kTile] = t_high[c];
            }
            if (wv == 0u) tile[2 * S * kTile + e] = 0u;  // (row 2S: the fold's look-ahead reads it)
        }
        wave::block_barrier();
        PHE_TILE_MARK(3);  // carries in, T written, barrier
        // ---- fold: this wave's C columns of y = T_low + sum_i T[P + i] * C_i ------------------------------------------------------
        uint32_t y_blk[CW];
        uint64_t y_carry;
        {
            const int c0 = (int)wv * CW;
            uint64_t acc[CW];
            typename TileUpper<CW>::word upper[CW];
#pragma unroll
            for (int k = 0; k < CW; ++k) {
                acc[k] = (c0 + k < P) ? t_low[k] : 0u;
                upper[k] = 0u;
            }
            // kFoldGroup fold digits per request group, two groups in flight: while one group is multiplied the table words
            // (scalar cache / L2 -> SGPRs) and digits (LDS) of the next one travel — 36 multiply-adds per wave, times the waves
            // of the SIMD, to cover an L2 round trip (the 84 KB table streams through a 16 KB scalar cache: every read
            // of it is an L2 read).  The table carries kFoldPadRows zero rows past the last digit for the look-ahead.
            const uint32_t* tw = A.table + (size_t)wv * (size_t)(D + kFoldPadRows) * CW;
            const uint32_t* digits = tile + (size_t)P * kTile + e;
            constexpr int GD = T::kFoldGroup, GP = GD / 2;
            wave::ScalarRow<CW> ca[GD], cb[GD];
            wave::DigitPair da[GP], db[GP];
            // the rows of request group G (0: the group at t / dg, 1 and 2: the next two) of the iteration whose first table row is t and
            // whose first digit row is dg: every offset an immediate of its load (round 6: sixteen waves share the CU's scalar unit, and a
            // pointer add per table row, a division for the priority and the ladder of branches behind it were ~100 scalar instructions
            // per 72 multiply-adds — the fold waited for the scalar unit, not for its multiply-adds)
            auto request = [&](auto group, wave::ScalarRow<CW> (&c)[GD], wave::DigitPair (&d)[GP], const uint32_t* t, const uint32_t* dg) __attribute__((always_inline)) {
                constexpr int G = decltype(group)::value;
                c[0].template request_at<(G * GD + 0) * CW>(t);
                c[1].template request_at<(G * GD + 1) * CW>(t);
                if constexpr (GD > 2) {
                    c[2].template request_at<(G * GD + 2) * CW>(t);
                    c[3].template request_at<(G * GD + 3) * CW>(t);
                }
                d[0].template request<G * GD>(dg);
                if constexpr (GP > 1) d[1].template request<G * GD + 2>(dg);
            };
            static_assert(GD == 2 || GD == 4, "request groups of two or four fold digits");
            auto multiply = [&](const wave::ScalarRow<CW> (&c)[GD], const wave::DigitPair (&d)[GP]) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < GD; ++u)
#pragma unroll
                    for (int k = 0; k < CW; ++k) acc[k] = wave::mad64(d[u / 2].word(u % 2), c[u].word(k), acc[k]);
            };
            using G0 = std::integral_constant<int, 0>;
            using G1 = std::integral_constant<int, 1>;
            using G2 = std::integral_constant<int, 2>;
            const uint32_t* t4 = tw;
            const uint32_t* d4 = digits;
            request(G0(), ca, da, t4, d4);
            const int n_it = (D + 2 * GD - 1) / (2 * GD), quarter = (n_it + 3) / 4;  // (D is a multiple of 8: whole iterations)
            int next_level = quarter, level = 3, since = 0;
#pragma unroll 1
            for (int it = 0; it < n_it; ++it) {
                if (it == next_level) {  // (three times a tile)
                    level -= 1;
                    wave::set_priority(level);
                    next_level += quarter;
                }
                wave::arrived<CW, GD>(ca, da);
                request(G1(), cb, db, t4, d4);
                multiply(ca, da);
                wave::arrived<CW, GD>(cb, db);
                request(G2(), ca, da, t4, d4);
                multiply(cb, db);
                t4 += 2 * GD * CW;
                d4 += 2 * GD * kTile;
                if (++since == kFoldChunk / (2 * GD)) {  // (kFoldChunk products of < 2^58.01 between two hand-overs)
                    tile_hand_over<CW>(acc, upper);
                    since = 0;
                }
            }
            wave::arrived<CW, GD>(ca, da);  // (the look-ahead past the last digit: nothing may still be travelling to an SGPR)
            PHE_TILE_MARK(4);  // fold
            y_carry = tile_block_carries<CW>(y_blk, acc, upper);
            if constexpr (!kSettleByBlocks) {
#pragma unroll
                for (int k = 0; k < CW; ++k) {
                    const int c = c0 + k;
                    if (c < P) tile[(size_t)e * P + c] = y_blk[k];
                    else top[e * kTableRowSlack + (c - P)] = y_blk[k];
                }
                fold_carry[(wv * kTile + e) * 2u] = (uint32_t)y_carry;
                fold_carry[(wv * kTile + e) * 2u + 1u] = (uint32_t)(y_carry >> 32);
            }
        }
        if constexpr (kSettleByBlocks) {
            // ---- settle: lane = element, wave = column block (tile_settle_blocks) ---------------------------------------------------
            const uint64_t raw_item = tile_i * kTile + e;
            const bool live = raw_item < A.batch;
            tile_settle_blocks<L, W>(A, tile, prod_carry, top, fold_carry, cst, wv, e, y_blk, y_carry, live ? raw_item : A.batch - 1, live,
                                     [&]() __attribute__((always_inline)) {
                                         PHE_TILE_MARK(5);
                                         request_rows(tile_i + n_blocks < n_tiles ? tile_i + n_blocks : tile_i);  // (see below)
                                     });
            PHE_TILE_MARK(6);
            wave::block_barrier();  // the digits are read: the next tile's may take the buffer
            PHE_TILE_MARK(7);
            continue;
        }
        wave::block_barrier();
        PHE_TILE_MARK(5);  // carries, columns to LDS, barrier
        // the next tile's rows travel under the settle (not under the fold: 64 rows per load instruction keep L2 busy, and the
        // fold lives on the latency of its table words from L2 — measured: the fold took 2.3 times as long with them in flight)
        // (requested for the last tile as well — its own rows again — so that the registers are dead from the cut to this point
        // on every path: a conditional request keeps the old values alive through the product and spills them)
        request_rows(tile_i + n_blocks < n_tiles ? tile_i + n_blocks : tile_i);
        // ---- settle: 16 lanes per element (lane g = columns [L g, L g + L)); a wave's 64 / W elements four at a time ----------------------
        constexpr int kPerWave = kTile / kTileWaves, kLanesPerBlock = CW / L, kAtOnce = kTile / GS;  // (elements a wave settles / at a time)
        static_assert(kPerWave == kAtOnce, "a wave settles its elements in one pass (GS = W)");
#pragma unroll 1
        for (int it = 0; it < kPerWave / kAtOnce; ++it) {
            const Lanes<GS> ln(lane);
            const uint32_t g = ln.g;
            const uint32_t es = wv * (uint32_t)kPerWave + (uint32_t)it * (uint32_t)kAtOnce + wave::reread(lane) / GS;
            uint32_t* row = tile + (size_t)T::kSettleRow0 * kTile + (wv * (uint32_t)kAtOnce + wave::reread(lane) / GS) * T::kRowT;
            const uint64_t raw_item = tile_i * kTile + es;
            const bool live = raw_item < A.batch;
            const uint64_t item = live ? raw_item : A.batch - 1;
            uint64_t acc[L];
            uint32_t t[L], n[L];
            {
                const uint32_t gi = wave::reread(g);
#pragma unroll
                for (int k = 0; k < L; ++k) {
                    const int c = (int)gi * L + k;
                    acc[k] = (c < P) ? tile[(size_t)es * P + c] : top[es * kTableRowSlack + (c - P)];
                }
                if (gi % kLanesPerBlock == 0u && gi >= (uint32_t)kLanesPerBlock) {  // the lane that opens block w': the carry of block w' - 1 enters
                    const uint32_t wb = gi / kLanesPerBlock - 1u;
                    acc[0] += ((uint64_t)fold_carry[(wb * kTile + es) * 2u + 1u] << 32) | fold_carry[(wb * kTile + es) * 2u];
                }
            }
            normalize_partial<GS, L>(t, acc, ln);  // (almost-normalised is enough for the estimate: limbs < 2^29 + a carry)
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
            {   // r = y - q^ N = (y + q^ (W^S - N)) mod W^S  <  3 N
                uint32_t c0[L], c1[L];
                load_row<L>(c0, cst + S, g);
                load_row<L>(c1, cst + 2 * S, g);
#pragma unroll
                for (int k = 0; k < L; ++k) acc[k] = wave::mad64(q1, c1[k], wave::mad64(q0, c0[k], (uint64_t)t[k]));
            }
            normalize_partial<GS, L>(t, acc, ln);  // (the carry out of the top lane — q^ itself — is the multiple of W^S dropped)
            load_row<L>(n, cst, g);
            canonicalize<GS, L>(t, n, ln);
            store_words<GS, L>(A.out + item * A.out_stride, A.limbs, t, row, wave::reread(g), live);
        }
        PHE_TILE_MARK(6);  // settle + store
        wave::block_barrier();  // the settled columns are read: the next tile's digits may take the buffer
        PHE_TILE_MARK(7);  // barrier
    }
#if defined(PHE_TILE_PROFILE)
    if (lane == 0u) {
        for (int i = 0; i < 8; ++i) atomicAdd((unsigned long long*)A.profile + i, (unsigned long long)prof_[i]);
        atomicAdd((unsigned long long*)A.profile + 8 + wv, (unsigned long long)prof_[1]);        // product clocks by wave
        atomicAdd((unsigned long long*)A.profile + 8 + 16 + wv, (unsigned long long)prof_[4]);   // fold clocks by wave
    }
#endif
}

}  // namespace phe
