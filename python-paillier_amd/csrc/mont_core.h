// mont_core.h — carry-free radix-2^29 Montgomery arithmetic for one limb group of G lanes.
//
// Replaces, for whole batches, what the reference delegates to gmpy2/libgmp one scalar at a time:
//   phe/util.py:38-50  powmod  -> modexp_uniform_body / modexp_var_body
//   phe/util.py:53-64  mulmod  -> mulmod_body
// and fuses the Paillier wrappers around them:
//   phe/paillier.py:102-139 raw_encrypt, :603-624 obfuscate, :328-354 raw_decrypt (the two CRT
//   half-exponentiations; the L/CRT tail is decrypt_tail.h), :705-719 _raw_add, :721-751 _raw_mul.
//
// Why radix 2^29.  Calibration on gfx950 (csrc/microbench.hip, profiles/microbench_r01.json): a
// v_mad_u64_u32 issues in ~4.8 cycles, but so does every instruction that touches a carry flag
// (v_add_co / v_addc_co ~4.4), and a full-radix (2^32) CIOS needs ~1.1 of those per multiply.  With
// 29-bit limbs held in 32-bit lanes the 58-bit products are summed directly in the 64-bit addend of
// v_mad_u64_u32: a column accumulator absorbs 2 products per row for as many rows as it stays in a
// lane (L <= 31 keeps it below 2^64 for any operands; L = 36 is used only for moduli that
// key_setup.h:accumulators_fit() has checked), so the inner loop is multiply-accumulates only.  The price is
// (ceil(bits/29)/ceil(bits/32))^2 ~ 1.27x more multiplies, a net ~1.5x fewer issue cycles.
//
// Layout.  A modulus N with 29*S >= bits(N) + 4 (S = G*L limbs) is owned by a group of G lanes
// (G = 16: one DPP row, G = 8: half a row, G = 4: one quad); lane g keeps limbs [g*L, (g+1)*L).  R = 2^(29*S) >= 16 N.
//
// montmul (word-serial, per 29-bit digit a_i of the multiplier, broadcast-read from LDS):
//     acc[k] += a_i * b[k]                      L  v_mad_u64_u32
//     m = (acc[0] * -N^-1) mod 2^29 of lane 0   v_mul_lo, v_and, DPP broadcast
//     acc[k] += m * n[k]                        L  v_mad_u64_u32
//     shift by one limb: the low 29 bits of acc[0] (zero in lane 0) move to the top of the lane below
//     (one DPP), the rest of acc[0] is added to acc[1]; accumulators rotate by renaming.
// Values are kept in [0, 2N) between products (R >= 16N makes that closed under montmul) with limbs
// "almost normalised" (< 2^29 + 2^8): one lane-local carry sweep and one neighbour hand-off per
// product, no lane-to-lane ripple, no comparison.  Only a kernel's final result is made canonical
// (full carry look-ahead over two ballots + conditional subtraction), which is what makes the output
// bit-identical to gmpy2's canonical residues.
//
// Written only against the `wave::` primitives (wave_gfx950.h on the device,
// tests/emu/wave_emu.h for CPU tests); otherwise plain C++.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace phe {

constexpr int kRadixBits = 29;
constexpr uint32_t kLimbMask = (1u << kRadixBits) - 1u;
constexpr int kLdsPad = 4;  // words of padding after each group's LDS operand (keeps 16-B alignment)

template <int G>
using Lanes = wave::Lanes<G>;

template <int G>
struct GroupMasks {
    static constexpr uint64_t lane0 = (G == 64) ? 0x0000000000000001ull
                                      : (G == 16) ? 0x0001000100010001ull
                                      : (G == 8) ? 0x0101010101010101ull
                                      : (G == 4) ? 0x1111111111111111ull
                                      : (G == 2) ? 0x5555555555555555ull
                                                 : 0xffffffffffffffffull;
    static constexpr uint64_t top = lane0 << (G - 1);
};

PHE_DEV uint32_t lane_bit(uint64_t mask, uint32_t lane) { return (uint32_t)(mask >> lane) & 1u; }
// kLimbMask in every lane, as plain VGPR data (so that "dpp(x) & mask" becomes one v_and_b32_dpp instead of v_and with a literal
// + v_mov_b32_dpp): no lane of a group of two or more is both its top and its low lane; a group of ONE lane (G = 1: one number
// per lane, no cross-lane step at all) is both
template <int G>
PHE_DEV uint32_t digit_mask(const Lanes<G>& ln) {
    if constexpr (G == 1) return wave::reread(kLimbMask);
    else return kLimbMask & (ln.not_top | ln.not_low);
}
template <int G>
PHE_DEV uint32_t group_top_bit(uint64_t mask, uint32_t lane) { return (uint32_t)(mask >> (lane | (G - 1))) & 1u; }

// Carry (or borrow) look-ahead across the lanes of every group at once.
//   gen  : lanes whose lane-local sweep produced a carry out
//   prop : lanes that would pass an incoming carry on (all limbs 2^29-1 / all-zero difference)
// gen and prop are disjoint by construction.  Returns the lanes that receive a carry-in;
// out_top gets (at the top-lane bit of each group) whether the group as a whole carried out.
template <int G>
PHE_DEV uint64_t group_carry_in(uint64_t gen, uint64_t prop, uint64_t& out_top) {
    const uint64_t gs = (gen << 1) & ~GroupMasks<G>::lane0;
    const uint64_t pm = prop & ~GroupMasks<G>::top;  // a top lane must not ripple into the next group's field
    const uint64_t cin = (gs + pm) ^ pm;
    out_top = (gen | (prop & cin)) & GroupMasks<G>::top;
    return cin;
}

// ---- normalisation ---------------------------------------------------------------------------
// 64-bit column sums -> almost-normalised 32-bit limbs (< 2^29 + 2^8), value unchanged.
template <int G, int L>
PHE_DEV void normalize_partial(uint32_t (&t)[L], const uint64_t (&acc)[L], const Lanes<G>& ln) {
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < L; ++k) {
        const uint64_t v = acc[k] + carry;
        t[k] = (uint32_t)v & kLimbMask;
        carry = v >> kRadixBits;
    }
    // the lane's carry (< 2^36) belongs to limb 0 of the lane above; the top lane's is 0 (value < R)
    const uint32_t inc_lo = wave::grp_up1<G>((uint32_t)carry, ln);
    const uint32_t inc_hi = wave::grp_up1<G>((uint32_t)(carry >> 32), ln);
    const uint64_t v = (uint64_t)t[0] + (((uint64_t)inc_hi << 32) | inc_lo);
    t[0] = (uint32_t)v & kLimbMask;
    const uint32_t c2 = (uint32_t)(v >> kRadixBits);  // < 2^8
    if constexpr (L > 1) {
        t[1] += c2;
    } else {
        t[0] += wave::grp_up1<G>(c2, ln);
    }
}

// t += u for two almost-normalised numbers; the sum is almost-normalised again (32-bit arithmetic only)
template <int G, int L>
PHE_DEV void add_normalize(uint32_t (&t)[L], const uint32_t (&u)[L], const Lanes<G>& ln) {
    uint32_t carry = 0;
#pragma unroll
    for (int k = 0; k < L; ++k) {
        const uint32_t v = t[k] + u[k] + carry;  // < 2^31
        t[k] = v & kLimbMask;
        carry = v >> kRadixBits;                 // <= 3
    }
    t[0] += wave::grp_up1<G>(carry, ln);         // the top lane's carry is 0 (value < R)
}

// almost-normalised -> canonical limbs (< 2^29) with all lane-to-lane carries resolved
template <int G, int L>
PHE_DEV void normalize_full(uint32_t (&t)[L], const Lanes<G>& ln) {
    uint32_t c = 0, ones = kLimbMask;
#pragma unroll
    for (int k = 0; k < L; ++k) {
        const uint32_t v = t[k] + c;
        t[k] = v & kLimbMask;
        c = v >> kRadixBits;
        ones &= t[k];
    }
    const uint64_t gen = wave::ballot(c != 0);
    const uint64_t prop = wave::ballot(ones == kLimbMask);
    uint64_t out_top;
    const uint64_t cin = group_carry_in<G>(gen, prop, out_top);
    uint32_t ci = lane_bit(cin, ln.lane);
#pragma unroll
    for (int k = 0; k < L; ++k) {
        const uint32_t v = t[k] + ci;
        t[k] = v & kLimbMask;
        ci = v >> kRadixBits;
    }
}

// canonical t <- t - N if t >= N
template <int G, int L>
PHE_DEV void cond_sub(uint32_t (&t)[L], const uint32_t (&n)[L], const Lanes<G>& ln) {
    uint32_t d[L];
    uint32_t br = 0, nz = 0;
#pragma unroll
    for (int k = 0; k < L; ++k) {
        const uint32_t v = t[k] - n[k] - br;  // in (-2^29, 2^29): the sign bit is the borrow
        br = v >> 31;
        d[k] = v & kLimbMask;
        nz |= d[k];
    }
    const uint64_t gen = wave::ballot(br != 0);
    const uint64_t prop = wave::ballot(nz == 0);
    uint64_t out_top;
    const uint64_t bin = group_carry_in<G>(gen, prop, out_top);
    const uint64_t take = ~out_top & GroupMasks<G>::top;  // no final borrow: t >= N
    uint32_t bi = lane_bit(bin, ln.lane);
    const uint32_t sel = group_top_bit<G>(take, ln.lane);
#pragma unroll
    for (int k = 0; k < L; ++k) {
        const uint32_t v = d[k] - bi;
        bi = v >> 31;
        t[k] = sel ? (v & kLimbMask) : t[k];
    }
}

// value in [0, 3N), almost-normalised  ->  the canonical residue in [0, N)
template <int G, int L>
PHE_DEV void canonicalize(uint32_t (&t)[L], const uint32_t (&n)[L], const Lanes<G>& ln) {
    normalize_full<G, L>(t, ln);
    cond_sub<G, L>(t, n, ln);
    cond_sub<G, L>(t, n, ln);
}

// ---- the product ---------------------------------------------------------------------------------
// out = a * b * R^-1 (mod N), out < 2N almost-normalised.
//   a : the group's S digits in LDS (almost-normalised, value < R);  b : registers, value < 2N
//   (any a*b < R*N is fine: a < R with b < N, or both < 2N).
// rows: digits of the multiplier that are swept, R = 2^(29 rows).  G*L for every geometry but the whole-wave one (G = 64),
// whose numbers need not fill the 64*L limbs of the wave: lanes beyond `rows` limbs hold zeros and stay zero.
template <int G, int L>
PHE_DEV void montmul(uint32_t (&out)[L], const uint32_t* a, const uint32_t (&b)[L], const uint32_t (&n)[L],
                     uint32_t n0inv, const Lanes<G>& ln, int rows = G * L) {
    const int S = rows;
    const uint32_t dmask = kLimbMask & ln.not_top;  // digit mask + "the top lane receives 0" as one v_and
    // kLimbMask as plain VGPR data (no lane of a group is both top and low): lets "dpp(x) & mask" be one v_and_b32_dpp
    const uint32_t vmask = digit_mask<G>(ln);
    uint64_t acc[L];
#pragma unroll
    for (int k = 0; k < L; ++k) acc[k] = 0;
#pragma unroll 1
    for (int i = 0; i < S; i += L) {
        // L digits per trip: the accumulators rotate by one slot per digit and are back in place after L
#pragma unroll
        for (int j = 0; j < L; ++j) {
            const uint32_t ai = a[i + j];
#pragma unroll
            for (int k = 0; k < L; ++k) acc[(k + j) % L] = wave::mad64(ai, b[k], acc[(k + j) % L]);
            const uint32_t m = wave::grp_bcast0<G>((uint32_t)acc[j] * n0inv, ln) & vmask;
#pragma unroll
            for (int k = 0; k < L; ++k) acc[(k + j) % L] = wave::mad64(m, n[k], acc[(k + j) % L]);
            const uint64_t low = acc[j];  // logical limb 0 (= 0 mod 2^29 in lane 0)
            const uint32_t recv = wave::grp_down1_raw<G>((uint32_t)low) & dmask;
            if constexpr (L > 1) {
                acc[(j + 1) % L] += low >> kRadixBits;
                acc[j] = recv;  // becomes logical limb L-1
            } else {
                acc[0] = (low >> kRadixBits) + recv;
            }
        }
    }
    normalize_partial<G, L>(out, acc, ln);
}

// ---- operand movement --------------------------------------------------------------------------------
// 29-bit limbs [first_limb + g*L, +L) of the little-endian 32-bit-word number at p (limbs32 words)
// (count: limbs of the chunk — positions at or beyond it read as zero; the default takes whatever the lanes cover)
template <int L>
PHE_DEV void load_u32_as_r29(uint32_t (&x)[L], const uint32_t* p, int limbs32, int first_limb, uint32_t g, int count = 1 << 30) {
    g = wave::reread(g);  // offsets recomputed here instead of hoisted out of the element loop (wave_gfx950.h:reread)
#pragma unroll
    for (int k = 0; k < L; ++k) {
        const int pos = (int)g * L + k;
        const int bit = kRadixBits * (first_limb + pos);
        const int q = bit >> 5, o = bit & 31;
        const uint64_t w0 = (q < limbs32 && pos < count) ? p[q] : 0u;
        const uint64_t w1 = (q + 1 < limbs32 && pos < count) ? p[q + 1] : 0u;
        x[k] = (uint32_t)(((w1 << 32) | w0) >> o) & kLimbMask;
    }
}
// full-width (S words) row of 29-bit limbs: window-table entries and per-modulus constants
template <int L>
PHE_DEV void load_row(uint32_t (&x)[L], const uint32_t* p, uint32_t g) {
#pragma unroll
    for (int k = 0; k < L; ++k) x[k] = p[g * L + k];
}
template <int L>
PHE_DEV void store_row(uint32_t* p, const uint32_t (&x)[L], uint32_t g) {
#pragma unroll
    for (int k = 0; k < L; ++k) p[g * L + k] = x[k];
}
// publish a value as the group's LDS multiplier operand
template <int L>
PHE_DEV void lds_put(uint32_t* row, const uint32_t (&x)[L], uint32_t g) {
    wave::lds_fence();
#pragma unroll
    for (int k = 0; k < L; ++k) row[g * L + k] = x[k];
    wave::lds_fence();
}
// canonical 29-bit limbs -> little-endian 32-bit words at p (limbs32 words), repacked through LDS so
// that consecutive lanes store consecutive words
template <int G, int L>
PHE_DEV void store_r29_as_u32(uint32_t* p, int limbs32, const uint32_t (&t)[L], uint32_t* row, uint32_t g,
                              bool live) {
    constexpr int S = G * L;
    g = wave::reread(g);
    lds_put<L>(row, t, g);
    if (live) {
        for (int j = (int)g; j < limbs32; j += G) {
            const int bit = 32 * j;
            const int q = bit / kRadixBits, o = bit - q * kRadixBits;
            uint64_t v = (q < S) ? row[q] : 0u;
            if (q + 1 < S) v |= (uint64_t)row[q + 1] << kRadixBits;
            if (q + 2 < S) v |= (uint64_t)row[q + 2] << (2 * kRadixBits);
            p[j] = (uint32_t)(v >> o);
        }
    }
    wave::lds_fence();
}

// ---- per-modulus constants (device pointers, S = G*L words of 29-bit limbs each) ------------------------
struct ModConsts {
    const uint32_t* n;    // N
    const uint32_t* r1;   // R   mod N  (Montgomery one)
    const uint32_t* r2;   // R^2 mod N
    const uint32_t* r3;   // R^3 mod N  (folds the part of a wide input above 2^(29*S))
    const uint32_t* aux;  // encrypt: n*R mod n^2, so montmul(m, aux) = n*m
    uint32_t n0inv;       // -N^-1 mod 2^29
};

enum : int { kModeEncrypt = 0, kModeObfuscate = 1, kModeHalfDecrypt = 2, kModePow = 3 };

// Batch-uniform exponent (encrypt/obfuscate: e = n;  decrypt halves: e = p-1, q-1).
// The host turns e into a sliding-window schedule; every op word is (squarings << 8) | (idx+1)
// where idx selects the odd power base^(2*idx+1) from the group's table (0 = no multiply).
// All user-visible numbers (base, post, out) are little-endian 32-bit-word rows.
struct UniformArgs {
    ModConsts mod;
    const uint32_t* sched;
    int n_ops;
    int first_idx;    // table entry the accumulator starts from (top window)
    int tbl_entries;  // 2^(w-1)
    const uint32_t* base;  // (batch, base_limbs): r | wide c | base
    int base_limbs;
    const uint32_t* post;  // encrypt: m (batch, post_limbs); obfuscate: c_in (batch, post_limbs)
    int post_limbs;
    uint32_t* out;  // (batch, out_limbs)
    int out_limbs;
    uint32_t* table;  // scratch: total_groups * tbl_entries * S words
    uint64_t batch;
};

template <int G, int L, int MODE>
PHE_DEV void modexp_uniform_body(const UniformArgs& A, uint32_t* lds_row, uint32_t slot, uint32_t total_slots,
                                 uint32_t lane) {
    constexpr int S = G * L;
    const Lanes<G> ln(lane);
    const uint32_t g = ln.g;
    const uint32_t n0inv = A.mod.n0inv;
    uint32_t n[L];
    load_row<L>(n, A.mod.n, g);
    uint32_t* tbl = A.table + (size_t)slot * (size_t)A.tbl_entries * S;
    // the wave iterates while any of its groups has work; an idle group recomputes the last item
    // harmlessly (all 64 lanes stay converged through the DPP/ballot steps)
    const uint64_t n_iter = (A.batch + total_slots - 1) / total_slots;
    for (uint64_t it = 0; it < n_iter; ++it) {
        uint64_t item = slot + it * (uint64_t)total_slots;
        const bool live = item < A.batch;
        if (!live) item = A.batch - 1;
        uint32_t acc[L], tmp[L], cst[L];
        // ---- base -> Montgomery form -----------------------------------------------------------
        const uint32_t* bp = A.base + item * (uint64_t)A.base_limbs;
        load_u32_as_r29<L>(tmp, bp, A.base_limbs, 0, g);
        lds_put<L>(lds_row, tmp, g);
        load_row<L>(cst, A.mod.r2, g);
        montmul<G, L>(acc, lds_row, cst, n, n0inv, ln);
        if (MODE == kModeHalfDecrypt) {
            // c = lo + hi*2^(29S)  ->  c*R = lo*R + hi*R^2; the sum (< 4N) is brought back below 2N by a
            // product with R mod N
            load_u32_as_r29<L>(tmp, bp, A.base_limbs, S, g);
            lds_put<L>(lds_row, tmp, g);
            load_row<L>(cst, A.mod.r3, g);
            montmul<G, L>(tmp, lds_row, cst, n, n0inv, ln);
            add_normalize<G, L>(tmp, acc, ln);  // tmp += acc, limbs brought back below 2^29 + 2^8
            lds_put<L>(lds_row, tmp, g);
            load_row<L>(cst, A.mod.r1, g);
            montmul<G, L>(acc, lds_row, cst, n, n0inv, ln);
        }
        // ---- odd powers base^1, base^3, ... in Montgomery form ----------------------------------
        store_row<L>(tbl, acc, g);
        if (A.tbl_entries > 1) {
            lds_put<L>(lds_row, acc, g);
            montmul<G, L>(cst, lds_row, acc, n, n0inv, ln);  // base^2
            for (int j = 1; j < A.tbl_entries; ++j) {
                lds_put<L>(lds_row, acc, g);
                montmul<G, L>(acc, lds_row, cst, n, n0inv, ln);
                store_row<L>(tbl + (size_t)j * S, acc, g);
            }
        }
        // ---- left-to-right sliding window ------------------------------------------------------
        load_row<L>(acc, tbl + (size_t)A.first_idx * S, g);
        for (int op = 0; op < A.n_ops; ++op) {
            const uint32_t w = A.sched[op];
            const int nsq = (int)(w >> 8);
            const int sel = (int)(w & 0xffu);
            for (int s = 0; s < nsq; ++s) {
                lds_put<L>(lds_row, acc, g);
                montmul<G, L>(acc, lds_row, acc, n, n0inv, ln);
            }
            if (sel) {
                lds_put<L>(lds_row, acc, g);
                load_row<L>(tmp, tbl + (size_t)(sel - 1) * S, g);
                montmul<G, L>(acc, lds_row, tmp, n, n0inv, ln);
            }
        }
        // ---- leave Montgomery form (fused with the op's final multiply) -------------------------
        if (MODE == kModeEncrypt) {
            // nude ciphertext 1 + n*m (phe/paillier.py:134; the :125-130 branch is value-identical)
            load_u32_as_r29<L>(tmp, A.post + item * (uint64_t)A.post_limbs, A.post_limbs, 0, g);
            lds_put<L>(lds_row, tmp, g);
            load_row<L>(cst, A.mod.aux, g);
            montmul<G, L>(tmp, lds_row, cst, n, n0inv, ln);  // n*m (mod n^2), < 2N
            if (g == 0u) tmp[0] += 1u;
        } else if (MODE == kModeObfuscate) {
            load_u32_as_r29<L>(tmp, A.post + item * (uint64_t)A.post_limbs, A.post_limbs, 0, g);
        } else {
#pragma unroll
            for (int k = 0; k < L; ++k) tmp[k] = (g == 0u && k == 0) ? 1u : 0u;
        }
        // acc = x*R, tmp = y (plain)  ->  montmul = x*y mod N, plain
        lds_put<L>(lds_row, tmp, g);
        montmul<G, L>(acc, lds_row, acc, n, n0inv, ln);
        canonicalize<G, L>(acc, n, ln);
        store_r29_as_u32<G, L>(A.out + item * (uint64_t)A.out_limbs, A.out_limbs, acc, lds_row, g, live);
    }
}

// Per-element exponents (phe/paillier.py:751 powmod(c, scalar, n^2); :749 with the inverted base).
// Fixed 2^w-ary windows over the batch-wide maximum bit length: every group runs the same
// schedule (no divergence inside a wave); the digit only picks the table entry.
struct VarArgs {
    ModConsts mod;
    const uint32_t* base;  // (batch, base_limbs), values < N
    int base_limbs;
    const uint32_t* exps;  // (batch, exp_limbs)
    int exp_limbs;
    int window;      // w in 1..6 (key_setup.h:pick_window)
    int n_windows;   // ceil(max_bits / w), >= 1
    uint32_t* out;
    int out_limbs;
    uint32_t* table;  // scratch: total_groups * 2^w * S words
    uint64_t batch;
};

PHE_DEV uint32_t exp_digit(const uint32_t* e, int exp_limbs, int bitpos, int w) {
    // bits [bitpos, bitpos+w) of e
    const int wi = bitpos >> 5, sh = bitpos & 31;
    uint64_t v = (wi < exp_limbs) ? e[wi] : 0u;
    if (wi + 1 < exp_limbs) v |= (uint64_t)e[wi + 1] << 32;
    return (uint32_t)(v >> sh) & ((1u << w) - 1u);
}

template <int G, int L>
PHE_DEV void modexp_var_body(const VarArgs& A, uint32_t* lds_row, uint32_t slot, uint32_t total_slots,
                             uint32_t lane) {
    constexpr int S = G * L;
    const Lanes<G> ln(lane);
    const uint32_t g = ln.g;
    const uint32_t n0inv = A.mod.n0inv;
    const int tbl_entries = 1 << A.window;
    uint32_t n[L];
    load_row<L>(n, A.mod.n, g);
    uint32_t* tbl = A.table + (size_t)slot * (size_t)tbl_entries * S;
    const uint64_t n_iter = (A.batch + total_slots - 1) / total_slots;
    for (uint64_t it = 0; it < n_iter; ++it) {
        uint64_t item = slot + it * (uint64_t)total_slots;
        const bool live = item < A.batch;
        if (!live) item = A.batch - 1;
        uint32_t acc[L], tmp[L], xm[L];
        load_u32_as_r29<L>(tmp, A.base + item * (uint64_t)A.base_limbs, A.base_limbs, 0, g);
        lds_put<L>(lds_row, tmp, g);
        load_row<L>(xm, A.mod.r2, g);
        montmul<G, L>(xm, lds_row, xm, n, n0inv, ln);  // base*R
        // table: base^0 .. base^(2^w - 1)
        load_row<L>(acc, A.mod.r1, g);
        store_row<L>(tbl, acc, g);
        store_row<L>(tbl + S, xm, g);
#pragma unroll
        for (int k = 0; k < L; ++k) acc[k] = xm[k];
        for (int j = 2; j < tbl_entries; ++j) {
            lds_put<L>(lds_row, acc, g);
            montmul<G, L>(acc, lds_row, xm, n, n0inv, ln);
            store_row<L>(tbl + (size_t)j * S, acc, g);
        }
        const uint32_t* e = A.exps + item * (uint64_t)A.exp_limbs;
        uint32_t d = exp_digit(e, A.exp_limbs, (A.n_windows - 1) * A.window, A.window);
        load_row<L>(acc, tbl + (size_t)d * S, g);
        for (int wi = A.n_windows - 2; wi >= 0; --wi) {
            for (int s = 0; s < A.window; ++s) {
                lds_put<L>(lds_row, acc, g);
                montmul<G, L>(acc, lds_row, acc, n, n0inv, ln);
            }
            d = exp_digit(e, A.exp_limbs, wi * A.window, A.window);
            if (wave::ballot(d != 0) != 0) {  // wave-uniform: skip when every group has a zero digit
                lds_put<L>(lds_row, acc, g);
                load_row<L>(tmp, tbl + (size_t)d * S, g);
                montmul<G, L>(acc, lds_row, tmp, n, n0inv, ln);
            }
        }
#pragma unroll
        for (int k = 0; k < L; ++k) tmp[k] = (g == 0u && k == 0) ? 1u : 0u;
        lds_put<L>(lds_row, tmp, g);
        montmul<G, L>(acc, lds_row, acc, n, n0inv, ln);
        canonicalize<G, L>(acc, n, ln);
        store_r29_as_u32<G, L>(A.out + item * (uint64_t)A.out_limbs, A.out_limbs, acc, lds_row, g, live);
    }
}

// out = a*b mod N  (phe/util.py:53-64 mulmod; phe/paillier.py:705-719 _raw_add).  a < R, b < N.
// Row strides are explicit so the same kernel walks the pair-product tree of batched inversion.
struct MulArgs {
    ModConsts mod;
    const uint32_t* a;
    const uint32_t* b;
    uint32_t* out;
    size_t a_stride, b_stride, out_stride;  // 32-bit words between consecutive rows
    int limbs;                              // 32-bit words per number
    int b_plain_limbs;                      // 0: b is a residue mod N.  > 0: b is a plaintext m of that many words and
                                            // the product is taken with 1 + n*m — E(a) * g^m, the "add a plaintext" of
                                            // EncryptedNumber._add_encoded (phe/paillier.py:673-675); needs mod.aux
    uint64_t batch;
    int a_limbs;                            // 0: rows of a hold `limbs` words; > 0: that many (a value up to R, e.g. a residue
                                            // modulo the scaled modulus n'^2 that is brought to n^2 by this product)
    int one_product;                        // mul_io.h: 1 = a*b*R^-1 mod N (one Montgomery product) instead of a*b mod N
    int vec_ok;                             // mul_io.h: every row pointer and stride is 16-byte aligned
};

template <int G, int L, bool ONE = false>
PHE_DEV void mulmod_body(const MulArgs& A, uint32_t* lds_row, uint32_t slot, uint32_t total_slots, uint32_t lane) {
    const Lanes<G> ln(lane);
    const uint32_t g = ln.g;
    const uint32_t n0inv = A.mod.n0inv;
    uint32_t n[L], r2[L];
    load_row<L>(n, A.mod.n, g);
    load_row<L>(r2, A.mod.r2, g);
    const uint64_t n_iter = (A.batch + total_slots - 1) / total_slots;
    for (uint64_t it = 0; it < n_iter; ++it) {
        uint64_t item = slot + it * (uint64_t)total_slots;
        const bool live = item < A.batch;
        if (!live) item = A.batch - 1;
        uint32_t x[L], y[L];
        load_u32_as_r29<L>(x, A.a + item * A.a_stride, A.a_limbs > 0 ? A.a_limbs : A.limbs, 0, g);
        if (A.b_plain_limbs > 0) {
            // nude ciphertext of the plaintext: 1 + n*m (mod n^2), value < 2N
            load_u32_as_r29<L>(y, A.b + item * A.b_stride, A.b_plain_limbs, 0, g);
            lds_put<L>(lds_row, y, g);
            load_row<L>(y, A.mod.aux, g);
            montmul<G, L>(y, lds_row, y, n, n0inv, ln);
            if (g == 0u) y[0] += 1u;
        } else {
            load_u32_as_r29<L>(y, A.b + item * A.b_stride, A.limbs, 0, g);
        }
        lds_put<L>(lds_row, x, g);
        montmul<G, L>(x, lds_row, y, n, n0inv, ln);   // a*b/R
        if constexpr (!ONE) {  // ONE: a*b*R^-1, one Montgomery product (A.one_product; see mul_io.h)
            lds_put<L>(lds_row, x, g);
            montmul<G, L>(x, lds_row, r2, n, n0inv, ln);  // a*b
        }
        canonicalize<G, L>(x, n, ln);
        store_r29_as_u32<G, L>(A.out + item * A.out_stride, A.limbs, x, lds_row, g, live);
    }
}

}  // namespace phe
