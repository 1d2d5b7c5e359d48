// mul_tile.h — a*b mod N (phe/util.py:53-64 mulmod; phe/paillier.py:705-719 _raw_add) as one plain product and one fold
// against the key's table (the arithmetic of mul_table.h), with the fold turned by 90 degrees: LANE = ELEMENT.
//
// mul_table.h folds inside the 16-lane limb group of the product: every lane reads its 9 table words per fold digit from LDS
// — 4 bytes of LDS per multiply-add, 256 B/clk asked of a CU whose LDS delivers 128 — and the 84 KB table sits in LDS next to
// everything else.  Measured (profiles/r04g): the fold half of the kernel runs at about a quarter of the multiply-add peak.
//
// Here a 512-thread workgroup takes a TILE of 64 elements through three phases:
//   1. product   (16-lane groups, mul_wide as before): T = a*b of element e leaves as 2S digits in column e of the tile
//                buffer  tile[row r][element e]  (r < S: the canonical low digits, r >= S: the high limbs);
//   2. fold      wave w owns the columns [2L w, 2L (w + 1)) of ALL 64 elements, one element per lane:
//                    y_c = lo_c + sum_i f_i * C_i[c]        f_i = tile[P + i][lane]   (one conflict-free LDS word per 2L products)
//                the table word C_i[c] is the same for every lane: it comes through the SCALAR data cache into an SGPR
//                (wave::scalar_words) and enters v_mad_u64_u32 as its scalar operand — no LDS, no VGPR, 4 bytes per 64
//                multiply-adds; the table (84 KB, [wave][digit][2L words]) stays in L2;
//   3. settle    (16-lane groups again) the column sums come back through LDS, then the quotient estimate, r = y - q^ N and the
//                conditional subtractions of mul_table.h.
// Between the phases the waves of the group meet at a barrier (4 per tile of 64 products, ~60 k clocks of work each).
// Same bits as mul_table.h, mul_io.h and gmpy2.mod(gmpy2.mul(a, b), c).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "mont_core.h"
#include "mul_io.h"
#include "mul_table.h"
#include "split_core.h"

// measurement-only build (tools/exp/tile_phases.hip): shader-clock time per phase, summed over the tiles of a wave
#if defined(PHE_TILE_PROFILE)
#define PHE_TILE_MARK(i)                                   \
    {                                                      \
        const uint64_t now_ = __builtin_readcyclecounter(); \
        prof_[i] += now_ - last_;                          \
        last_ = now_;                                      \
    }
#else
#define PHE_TILE_MARK(i) ((void)0)
#endif

namespace phe {

constexpr int kTile = 64;       // elements of a tile = lanes of a wave
constexpr int kTileWaves = 8;   // waves of the workgroup = column blocks of the fold
constexpr int kFoldPadRows = 4; // zero rows the column-block table carries past its last digit (the fold's look-ahead)
constexpr int kFoldChunk = 48;  // fold digits between two hand-overs of the accumulators' upper halves (48 * 2^58.01 + 2^32 < 2^64)

// LDS words of the workgroup: tile buffer | top columns | block carries | n, ncomp, ncomp1 | 32 digit rows | 8 waves x (stage a | b)
template <int L>
constexpr int tile_lds_words() {
    return 2 * 16 * L * kTile + kTile * kTableRowSlack + kTile * kTileWaves * 2 + 3 * 16 * L + 32 * (16 * L + kTableRowSlack) +
           kTileWaves * 2 * RowIO<16, L>::kStageWave;
}

// A.table: the fold table in the COLUMN-BLOCK layout [wave w][digit i][2L words]: limbs [2L w, 2L (w + 1)) of W^(P+i) mod N
// (key_setup.h:build_table_mul writes both layouts).  tile: 2 S kTile words; top: kTile * kTableRowSlack; carries: kTile *
// kTileWaves * 2; row: the group's digit row (S + kTableRowSlack); stage: the wave's staging area; cst: n | ncomp | ncomp1.
// `wv` must be wave-uniform.  Every wave of the workgroup runs the same number of tiles (the barriers).
template <int L>
PHE_DEV void mul_tile_body(const TableMulArgs& A, uint32_t* tile, uint32_t* top, uint32_t* carries, uint32_t* row, uint32_t* stage,
                           const uint32_t* cst, uint32_t wv, uint32_t block, uint32_t n_blocks, uint32_t lane) {
    constexpr int G = 16, S = G * L, CW = 2 * L;
    using IO = RowIO<G, L>;
    const Lanes<G> ln(lane);
    const uint32_t g = ln.g, gw = lane / G;
    uint32_t* stage_a = stage;
    uint32_t* stage_b = stage + IO::kStageWave;
    const int P = A.split, D = A.digits, n_lo = S - P;
    PHE_BOUNDS(n_lo >= 2 && n_lo <= kTableRowSlack && P + D <= 2 * S && A.base >= 0 && A.base + 3 < S && wv < (uint32_t)kTileWaves);
#pragma unroll
    for (int t = 0; t < 2 * IO::kVec; ++t) {  // chunks at or beyond the row length are never copied: they must read as zero
        Words4 z;
        z.x = z.y = z.z = z.w = 0u;
        *reinterpret_cast<Words4*>(stage + t * 256 + 4 * (int)lane) = z;
    }
    wave::lds_fence();
    const uint64_t n_tiles = (A.batch + kTile - 1) / kTile;
    // element of the tile this limb group works on in half `it` of phases 1 and 3, and its row of the batch (clamped)
    auto element = [&](int it) { return wv * 8u + (uint32_t)it * 4u + wave::reread(gw); };
    auto item_of = [&](uint64_t tile_i, int it) {
        const uint64_t item = tile_i * kTile + element(it);
        return item < A.batch ? item : A.batch - 1;
    };
    if (block < n_tiles) {
        const uint64_t first = item_of(block, 0);
        stage_row_async<G, L>(stage_a, A.a + first * A.a_stride, A.limbs, g);
        stage_row_async<G, L>(stage_b, A.b + first * A.b_stride, A.limbs, g);
    }
#if defined(PHE_TILE_PROFILE)
    uint64_t prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_readcyclecounter();
#endif
    for (uint64_t tile_i = block; tile_i < n_tiles; tile_i += n_blocks) {
        // ---- phase 1: T = a*b, 2S digits into column e of the tile buffer ---------------------------------------------------------
#pragma unroll 1
        for (int it = 0; it < 2; ++it) {
            const uint32_t e = element(it);
            uint32_t y[L], hi[L], zero[L];
            wave::wait_async_copies();
            {
                const uint32_t gi = wave::reread(g), gwi = wave::reread(gw);
                limbs_from_stage<G, L>(y, stage_b, gwi, gi);
                digits_from_stage<G, L>(row, stage_a, gwi, gi);
            }
            wave::lds_fence();
            {   // the next rows of this group: copied while this product runs (and, after the second half, under phases 2 and 3)
                const bool more = it == 0 || tile_i + n_blocks < n_tiles;
                if (more) {
                    const uint64_t nxt = it == 0 ? item_of(tile_i, 1) : item_of(tile_i + n_blocks, 0);
                    const uint32_t gi = wave::reread(g);
                    stage_row_async<G, L>(stage_a, A.a + nxt * A.a_stride, A.limbs, gi);
                    stage_row_async<G, L>(stage_b, A.b + nxt * A.b_stride, A.limbs, gi);
                }
            }
#pragma unroll
            for (int k = 0; k < L; ++k) zero[k] = 0u;
            mul_wide<G, L>(hi, row, y, zero, tile + e, ln, S, kTile);
            {
                const uint32_t gi = wave::reread(g);
#pragma unroll
                for (int k = 0; k < L; ++k) tile[(S + (int)gi * L + k) * kTile + (int)e] = hi[k];
            }
        }
        PHE_TILE_MARK(0);  // product
        wave::block_barrier();
        PHE_TILE_MARK(1);  // barrier
        // ---- phase 2: lane = element; this wave's 2L columns of y = lo_kept + sum_i f_i * C_i ------------------------------------
        {
            const uint32_t e = wave::reread(lane);
            const int c0 = (int)wv * CW;
            uint64_t acc[CW];
            uint64_t upper[CW];  // what the accumulators' upper halves held at the hand-overs: column = acc + upper * 2^32
                                 // (D * 2^58 can pass 2^65: the sum of the upper halves does not fit 32 bits)
#pragma unroll
            for (int k = 0; k < CW; ++k) {
                acc[k] = (c0 + k < P) ? tile[(c0 + k) * kTile + (int)e] : 0u;
                upper[k] = 0u;
            }
            wave::block_barrier();  // (the settled columns below land on the rows the other waves open their accumulators from)
            PHE_TILE_MARK(2);  // accumulators opened + barrier
            // Two fold digits per request group, two groups in flight: while one group is multiplied the table words (scalar
            // cache / L2 -> SGPRs) and digits (LDS) of the next one travel — 2 x 2L multiply-adds per wave, twice that with the
            // SIMD's other wave, to cover an L2 round trip (the 84 KB table streams through a 16 KB scalar cache: every read
            // of it is an L2 read).  The table carries kFoldPadRows zero rows past the last digit for the look-ahead.
            const uint32_t* tw = A.table + (size_t)wv * (size_t)(A.digits_padded + kFoldPadRows) * CW;
            const uint32_t* digits = tile + (size_t)P * kTile + e;
            wave::ScalarRow<CW> ca[2], cb[2];
            wave::DigitPair da, db;
            auto multiply = [&](const wave::ScalarRow<CW> (&c)[2], const wave::DigitPair& d) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int k = 0; k < CW; ++k) acc[k] = wave::mad64(d.word(u), c[u].word(k), acc[k]);
            };
            ca[0].request(tw);
            ca[1].request(tw + CW);
            da.template request<0>(digits);
#pragma unroll 1
            for (int i0 = 0; i0 < A.digits_padded; i0 += kFoldChunk) {
                const int n = (A.digits_padded - i0 < kFoldChunk) ? A.digits_padded - i0 : kFoldChunk;
#pragma unroll 1
                for (int i = 0; i < n; i += 4) {
                    const uint32_t* t4 = tw + (size_t)(i0 + i) * CW;
                    const uint32_t* d4 = digits + (size_t)(i0 + i) * kTile;
                    wave::arrived(ca[0], ca[1], da);
                    cb[0].request(t4 + 2 * CW);
                    cb[1].request(t4 + 3 * CW);
                    db.template request<2>(d4);
                    multiply(ca, da);
                    wave::arrived(cb[0], cb[1], db);
                    ca[0].request(t4 + 4 * CW);
                    ca[1].request(t4 + 5 * CW);
                    da.template request<4>(d4);
                    multiply(cb, db);
                }
#pragma unroll
                for (int k = 0; k < CW; ++k) {
                    upper[k] += acc[k] >> 32;
                    acc[k] &= 0xffffffffull;
                }
            }
            PHE_TILE_MARK(3);  // fold
            wave::arrived(ca[0], ca[1], da);  // (the look-ahead past the last digit: nothing may still be travelling to an SGPR)
            // the carries run inside the lane; what leaves the block (< 2^38) is added by the settle phase one column up
            uint64_t carry = 0;
#pragma unroll
            for (int k = 0; k < CW; ++k) {
                const uint64_t v = acc[k] + carry;
                const uint32_t digit = (uint32_t)v & kLimbMask;
                carry = (v >> kRadixBits) + (upper[k] << (32 - kRadixBits));
                const int c = c0 + k;
                if (c < P) tile[(size_t)e * P + c] = digit;
                else top[e * kTableRowSlack + (c - P)] = digit;
            }
            carries[(wv * kTile + e) * 2u] = (uint32_t)carry;
            carries[(wv * kTile + e) * 2u + 1u] = (uint32_t)(carry >> 32);
        }
        PHE_TILE_MARK(4);  // carries, columns to LDS
        wave::block_barrier();
        PHE_TILE_MARK(5);  // barrier
        // ---- phase 3: back on the limb groups: y canonical, q^ = floor(y / N) - 1 or - 2, r = y - q^ N < 3 N, the residue ---------
#pragma unroll 1
        for (int it = 0; it < 2; ++it) {
            const uint32_t e = element(it);
            const uint64_t raw_item = tile_i * kTile + e;
            const bool live = raw_item < A.batch;
            const uint64_t item = live ? raw_item : A.batch - 1;
            uint64_t acc[L];
            uint32_t t[L], n[L];
            {
                const uint32_t gi = wave::reread(g);
#pragma unroll
                for (int k = 0; k < L; ++k) {
                    const int c = (int)gi * L + k;
                    acc[k] = (c < P) ? tile[(size_t)e * P + c] : top[e * kTableRowSlack + (c - P)];
                }
                if ((gi & 1u) == 0u && gi >= 2u) {  // column 2L w' opens block w': the carry of block w' - 1 enters here
                    const uint32_t wb = gi / 2u - 1u;
                    acc[0] += ((uint64_t)carries[(wb * kTile + e) * 2u + 1u] << 32) | carries[(wb * kTile + e) * 2u];
                }
            }
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
            {
                uint32_t c0[L], c1[L];
                load_row<L>(c0, cst + S, g);
                load_row<L>(c1, cst + 2 * S, g);
#pragma unroll
                for (int k = 0; k < L; ++k) acc[k] = wave::mad64(q1, c1[k], wave::mad64(q0, c0[k], (uint64_t)t[k]));
            }
            normalize_partial<G, L>(t, acc, ln);  // (the carry out of the top lane — q^ itself — is the multiple of W^S dropped)
            load_row<L>(n, cst, g);
            canonicalize<G, L>(t, n, ln);
            store_words<G, L>(A.out + item * A.out_stride, A.limbs, t, row, wave::reread(g), live);
        }
        PHE_TILE_MARK(6);  // settle + store
        wave::block_barrier();  // the settled columns are read: the next tile's products may take the buffer
        PHE_TILE_MARK(7);  // barrier
    }
#if defined(PHE_TILE_PROFILE)
    if (lane == 0u)
        for (int i = 0; i < 8; ++i) atomicAdd((unsigned long long*)A.profile + i, (unsigned long long)prof_[i]);
#endif
}

}  // namespace phe
