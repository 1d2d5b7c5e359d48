// mul_io.h — the element-wise product kernels (phe/paillier.py:705-719 _raw_add, :673-675 add a plaintext, the product
// trees of batched inversion and of the multi-exponentiations) with their operand traffic made explicit.
//
// One product is ~40 k multiply-adds against 1.5 KB moved: the only kernel family of the path where the loads and
// stores are a visible share of the time.  Round 1's mulmod_body read every 29-bit limb with its own pair of 4-byte
// global loads (lane g of a group starts at bit 29*g*L: 64 different cache lines per wave instruction, 72 such
// instructions per element) and waited for them before the first multiply.  Here
//   * rows travel as 16-byte chunks, consecutive lanes of a limb group on consecutive chunks: in by LDS-DMA
//     (global_load_lds_dwordx4: global -> LDS without a register round trip; the destination is lane-linear, so the
//     staging area is laid out [chunk][lane][4 words] per wave), out by global_store_dwordx4; the 32-bit words are
//     re-sliced into 29-bit limbs (and back) from LDS;
//   * the copies of element i+1 are issued right after the limbs of element i have been sliced out, so they land under
//     the ~20-40 k multiply-adds of element i instead of being waited for (a register prefetch was tried first: the
//     allocator spilled the prefetched words, which made the kernel wait for the loads before the products after all);
//   * `one_product` gives a*b*R^-1 mod N (canonical) — ONE Montgomery product.  Resident ciphertext vectors use it for
//     chains of homomorphic additions: the missing powers of R are tracked per vector and settled by a single
//     product with R^(d+1) mod N when the plain residues are needed (phe/ciphertext.py, "Montgomery debt").
//
// Same MulArgs as mont_core.h:mulmod_body, which stays as the reference form (tests compare the two).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "mont_core.h"

namespace phe {

template <int G, int L>
struct RowIO {
    static constexpr int S = G * L;
    static constexpr int kWords = (S * kRadixBits + 31) / 32;           // 32-bit words that cover S limbs
    static constexpr int kVec = (kWords + 1 + 4 * G - 1) / (4 * G);     // 16-byte chunks per lane (+1 word: the window of the top limb)
    static constexpr int kRow = (S + kLdsPad + 3) & ~3;                 // a group's digit row, padded to 16 bytes
    // staging area of ONE WAVE for one operand: chunk t of every lane, lane-linear (what the LDS-DMA writes):
    // word (t, lane, i) at t*256 + 4*lane + i
    static constexpr int kStageWave = 256 * kVec;
    // LDS words of a 256-thread workgroup: digit rows of its 256/G groups | 4 waves x (stage a | stage b)
    static constexpr int kRowsWords = (256 / G) * kRow;
    static constexpr int kConstWords = (S + 3) & ~3;                    // R^2 mod N, shared by the workgroup
    static constexpr int kBlockWords = kRowsWords + 4 * 2 * kStageWave + kConstWords;
    static constexpr bool kUse = kBlockWords * 4 <= 65536 && L <= 18;
};

// word q of the row of group `gw` (its index inside the wave) in a wave's staging area
template <int G>
PHE_DEV uint32_t stage_word(const uint32_t* stage, uint32_t gw, int q) {
    constexpr int kChunk = 4 * G;  // words of one group per chunk index
    return stage[(q / kChunk) * 256 + (int)gw * kChunk + (q % kChunk)];
}
// start the copies of one row (limbs32 words, 16-byte aligned, limbs32 % 4 == 0) into the wave's staging area
template <int G, int L>
PHE_DEV void stage_row_async(uint32_t* stage, const uint32_t* p, int limbs32, uint32_t g) {
    PHE_BOUNDS(limbs32 <= RowIO<G, L>::kVec * 4 * G);  // every word of the row has a chunk to land in
#pragma unroll
    for (int t = 0; t < RowIO<G, L>::kVec; ++t) {
        const int w = 4 * (t * G + (int)g);
        wave::async_copy16_to_lds(p + w, stage + t * 256, w < limbs32);
    }
}
// 29-bit limbs [g*L, (g+1)*L) of the staged number
template <int G, int L>
PHE_DEV void limbs_from_stage(uint32_t (&x)[L], const uint32_t* stage, uint32_t gw, uint32_t g) {
#pragma unroll
    for (int k = 0; k < L; ++k) {
        const int bit = kRadixBits * ((int)g * L + k);
        const int q = bit >> 5, o = bit & 31;
        PHE_BOUNDS(q + 1 < RowIO<G, L>::kVec * 4 * G && (int)gw < 64 / G);  // both words lie in the chunks the staging area holds
        const uint64_t v = ((uint64_t)stage_word<G>(stage, gw, q + 1) << 32) | stage_word<G>(stage, gw, q);
        x[k] = (uint32_t)(v >> o) & kLimbMask;
    }
}
// the same, written straight to the group's digit row (the multiplier of the next product): no registers held
template <int G, int L>
PHE_DEV void digits_from_stage(uint32_t* row, const uint32_t* stage, uint32_t gw, uint32_t g) {
#pragma unroll
    for (int k = 0; k < L; ++k) {
        const int bit = kRadixBits * ((int)g * L + k);
        const int q = bit >> 5, o = bit & 31;
        PHE_BOUNDS(q + 1 < RowIO<G, L>::kVec * 4 * G && (int)g * L + k < RowIO<G, L>::kRow);
        const uint64_t v = ((uint64_t)stage_word<G>(stage, gw, q + 1) << 32) | stage_word<G>(stage, gw, q);
        row[(int)g * L + k] = (uint32_t)(v >> o) & kLimbMask;
    }
}

struct alignas(16) Words4 {
    uint32_t x, y, z, w;
};
// canonical limbs -> 32-bit words at p, as 16-byte chunks; `row` is the group's digit row (scratch)
template <int G, int L>
PHE_DEV void store_words(uint32_t* p, int limbs32, const uint32_t (&t)[L], uint32_t* row, uint32_t g, bool live) {
    constexpr int S = G * L;
    lds_put<L>(row, t, g);
    if (live) {
#pragma unroll
        for (int c = 0; c < RowIO<G, L>::kVec; ++c) {
            const int w = 4 * (c * G + (int)g);
            if (w < limbs32) {
                uint32_t o4[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int bit = 32 * (w + i);
                    const int q = bit / kRadixBits, o = bit - q * kRadixBits;
                    uint64_t v = (q < S) ? row[q] : 0u;
                    if (q + 1 < S) v |= (uint64_t)row[q + 1] << kRadixBits;
                    if (q + 2 < S) v |= (uint64_t)row[q + 2] << (2 * kRadixBits);
                    o4[i] = (uint32_t)(v >> o);
                }
                Words4 out;
                out.x = o4[0]; out.y = o4[1]; out.z = o4[2]; out.w = o4[3];
                *reinterpret_cast<Words4*>(p + w) = out;
            }
        }
    }
    wave::lds_fence();
}

// out = a*b mod N (two products), or a*b*R^-1 mod N (ONE = A.one_product), canonical.  b_stride == 0: one row b for the
// whole batch (a constant such as R^(d+1) mod N).  Needs A.vec_ok (16-byte aligned rows, limbs a multiple of 4): the
// launcher takes mont_core.h:mulmod_body otherwise.
//   row   : the group's digit row (RowIO::kRow words)
//   stage : the WAVE's staging area, 2 * RowIO::kStageWave words (a | b), 16-byte aligned
//   r2_row: R^2 mod N as S limbs in LDS (shared by the workgroup; written before the call)
template <int G, int L, bool ONE>
PHE_DEV void mul_io_body(const MulArgs& A, uint32_t* row, uint32_t* stage, uint32_t* r2_row, uint32_t slot,
                         uint32_t total_slots, uint32_t lane) {
    using IO = RowIO<G, L>;
    const Lanes<G> ln(lane);
    const uint32_t g = ln.g, gw = lane / G;
    const uint32_t n0inv = A.mod.n0inv;
    uint32_t* stage_a = stage;
    uint32_t* stage_b = stage + IO::kStageWave;
    const int b_limbs = A.b_plain_limbs > 0 ? A.b_plain_limbs : A.limbs;
    const int a_limbs = A.a_limbs > 0 ? A.a_limbs : A.limbs;
    uint32_t n[L];
    load_row<L>(n, A.mod.n, g);
    // chunks at or beyond the row length are never copied: they must read as zero
#pragma unroll
    for (int t = 0; t < 2 * IO::kVec; ++t) {
        Words4 z;
        z.x = z.y = z.z = z.w = 0u;
        *reinterpret_cast<Words4*>(stage + t * 256 + 4 * (int)lane) = z;
    }
    wave::lds_fence();
    const uint64_t n_iter = (A.batch + total_slots - 1) / total_slots;
    {
        const uint64_t first = (slot < A.batch) ? slot : A.batch - 1;
        stage_row_async<G, L>(stage_a, A.a + first * A.a_stride, a_limbs, g);
        stage_row_async<G, L>(stage_b, A.b + first * A.b_stride, b_limbs, g);
    }
    for (uint64_t it = 0; it < n_iter; ++it) {
        uint64_t item = slot + it * (uint64_t)total_slots;
        const bool live = item < A.batch;
        if (!live) item = A.batch - 1;
        uint32_t x[L], y[L];
        wave::wait_async_copies();
        {
            const uint32_t gi = wave::reread(g), gwi = wave::reread(gw);
            limbs_from_stage<G, L>(y, stage_b, gwi, gi);
            digits_from_stage<G, L>(row, stage_a, gwi, gi);
        }
        wave::lds_fence();
        if (it + 1 < n_iter) {  // the next element's rows: copied while this one is multiplied
            uint64_t nxt = slot + (it + 1) * (uint64_t)total_slots;
            if (nxt >= A.batch) nxt = A.batch - 1;
            const uint32_t gi = wave::reread(g);
            stage_row_async<G, L>(stage_a, A.a + nxt * A.a_stride, a_limbs, gi);
            stage_row_async<G, L>(stage_b, A.b + nxt * A.b_stride, b_limbs, gi);
        }
        if (A.b_plain_limbs > 0) {
            // nude ciphertext of the plaintext: 1 + n*m (mod n^2), value < 2N; the plaintext is the multiplier
            // (digits), then a's digits take the row
            uint32_t am[L];
            load_row<L>(am, row, g);                // a's digits back from the row (rare path: add a plaintext)
            lds_put<L>(row, y, g);
            load_row<L>(y, wave::reread_ptr(A.mod.aux), g);
            montmul<G, L>(y, row, y, n, n0inv, ln);
            if (g == 0u) y[0] += 1u;
            lds_put<L>(row, am, g);
        }
        montmul<G, L>(x, row, y, n, n0inv, ln);  // a*b/R
        if constexpr (!ONE) {
            load_row<L>(y, r2_row, g);  // R^2 mod N from LDS: a global load here would wait for the copies in flight too
            lds_put<L>(row, x, g);
            montmul<G, L>(x, row, y, n, n0inv, ln);  // a*b
        }
        canonicalize<G, L>(x, n, ln);
        store_words<G, L>(A.out + item * A.out_stride, A.limbs, x, row, wave::reread(g), live);
    }
}

}  // namespace phe
