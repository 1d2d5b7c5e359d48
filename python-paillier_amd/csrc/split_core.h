// split_core.h — arithmetic modulo n^2 on pairs of half-width numbers ("n-adic" split of the modulus).
//
// The moduli of Paillier are squares: n^2 for encrypt/obfuscate (phe/paillier.py:137, :622 powmod(r, n, nsquare)),
// p^2 and q^2 for the CRT halves of decrypt (:347, :351).  mont_core.h treats them as opaque 2k-bit moduli; here
// an element x of Z/n^2 is kept in HALF-width Montgomery form, x~ = x*R mod n^2 with R = 2^(29 H) >= 16 n, written
// in n-adic digits with a minus sign:
//
//        x~ = X0 - n * X1   (mod n^2),      X0, X1 numbers modulo n (H limbs each).
//
// With MQ a Montgomery product modulo n that keeps its quotient m (X0*Y0 + m*n = u*R exactly),
// x~ y~ = X0 Y0 - n (X0 Y1 + X1 Y0) = u R - n (m + X0 Y1 + X1 Y0)  (mod n^2), so
//
//     x*y :  Z0 = u,   Z1 = (m + X0*Y1 + X1*Y0) * R^-1 mod n
//     x^2 :  Z0 = u,   Z1 = (m + X0*(2 X1))     * R^-1 mod n
//
// Both words come out of ONE sweep over the H digits of X0 (pair_pass2 / pair_pass3): two column-accumulator sets
// advance together, and the quotient digit m_i of the first enters the second at the column it belongs to, so the
// quotient never exists as a number.  In multiply-adds per lane a squaring costs 4*L*H and a product 5*L*H, against
// 8*L*H for either on the full-width modulus (2L limbs per lane, 2H digits): the exponentiations of encrypt and
// decrypt are ~85 % squarings, so about half of the multiply-adds disappear.
// tools/exp/split_model.py checks the algebra and the lazy-reduction bounds below with plain integers.
//
// Lazy bounds (R >= 16 n): X0 < 2n, X1 < 2n are closed under both operations (X1 < 3n where a product is made of
// single sweeps, see split_mul; inputs up to 3n are fine everywhere); sums made while converting inputs stay below R.  Limbs are almost-normalised (< 2^29 + 2^8) exactly as in mont_core.h, and
// a column accumulator takes at most three products per digit for L digits: L <= 21 keeps it below 2^64 for any
// operands.
//
// Conversions happen once per element: in  — x = sum_j x_j R^j (x_j < R) is the sum of the pair products
// (x_j, 0) * pair(R^(j+2));  out — with (u, m) = MQ(X0, 1):  x*(1 + n*mp) = u + n*t,
// t = (mp*X0 - X1 - m) * R^-1 mod n = MONT(mp, X0) + MAC2(X1, n-1; m, n-1), made canonical.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <type_traits>

#include "mont_core.h"

namespace phe {

// A fused sweep (pair_pass2 / pair_pass3) keeps two accumulator sets and three operand rows in registers (7L VGPRs)
// and adds up to three products per digit to one accumulator: both stop at about L = 21 limbs per lane.  Wider lanes
// run every word of a pair product as its own single-accumulator sweep (two products per digit, L <= 31).
constexpr int kMaxFusedL = 21;
// ... of which the sweeps with TWO products per digit and accumulator (pair_pass2: squarings and the conversions in) have room for
// more by the arithmetic: 2 L products of < 2^58.0000014 and the per-row carries stay below 2^64 up to L = 31.  Round 6 tried them
// fused on the L = 27 rungs (3072-bit keys: n on 4 x 27, p and q on 2 x 27) — PHE_VARIANT_WIDE_FUSED, measurement only: bit-exact and
// FOURTEEN times slower (12.9 k against 182.7 k encrypts/s, same box: profiles/r06b_wide_rungs_fused_squarings.txt): two accumulator
// sets of 27 columns and three operand rows are 7 L = 189 registers before a single temporary, and the 256-register sweep loop
// spills every row.  The wide rungs stay on single sweeps.
#if defined(PHE_VARIANT_WIDE_FUSED)
constexpr int kMaxFusedPass2L = 27;
#else
constexpr int kMaxFusedPass2L = kMaxFusedL;
#endif
static_assert(kMaxFusedPass2L <= 31, "2 L products of < 2^58.0000014 per accumulator and stay must stay below 2^64");

// per-modulus constants (device pointers; H = G*L words of 29-bit limbs per row)
struct SplitConsts {
    const uint32_t* n;     // n
    const uint32_t* r1;    // R mod n
    const uint32_t* e;     // pair of 1: E0 | E1              (R mod n^2 = E0 - n*E1)
    const uint32_t* conv;  // chunk j: D0_j | D1_j             (R^(j+2) mod n^2 = D0 - n*D1)
    const uint32_t* nsq;   // n^2, 2H limbs
    uint32_t n0inv;        // -n^-1 mod 2^29
    int rows;              // limbs of a number (= G*L, except on the whole-wave geometry G = 64: key_setup.h SplitPack::rows)
    // wave-pair kernels only (key_setup.h QuickPack), null elsewhere:
    const uint32_t* nbar;  // scaled constants: (n~ + 1) / 2^29 for n~ = k*n = -1 (mod 2^29), H limbs
    const uint32_t* kx;    // exit constants: k*(n - 1) mod n, H limbs (the second word of a pair modulo n~ is X1/k modulo n)
};

// Same contract as UniformArgs (mont_core.h): batch-uniform exponent given as a sliding-window schedule,
// all user-visible numbers are little-endian 32-bit-word rows.
struct SplitArgs {
    SplitConsts mod;
    SplitConsts exit_mod;  // wave-pair kernels: `mod` is the scaled modulus n~, this one the true n with the same R (the way out)
    const uint32_t* sched;
    int n_ops;
    int first_idx;
    int tbl_entries;
    const uint32_t* base;  // (batch, base_limbs): r | wide c
    int base_limbs;
    int base_chunks;       // ceil(32*base_limbs / (29 H))
    const uint32_t* post;  // encrypt: m (batch, post_limbs); obfuscate: c_in (batch, post_limbs)
    int post_limbs;
    int post_chunks;
    uint32_t* out;  // (batch, out_limbs)
    int out_limbs;
    uint32_t* table;  // scratch: total_groups * tbl_entries * 2H words (entry = X0 row | X1 row)
    uint64_t batch;
    // wave-pair kernels: nullptr = one schedule for the batch; else four words per number {n_ops, first_idx, tbl_entries, offset of
    // its ops in `sched`} — per-element exponents (_raw_mul of a handful of numbers, phe/paillier.py:751), each with its own
    // sliding-window schedule made on the host (phe_hip_powmod)
    const uint32_t* item_meta;
};

// ---- squarings: the symmetric half of X0*X0 ---------------------------------------------------------------------
// A squaring sweep multiplies the digits of X0 (LDS) into the lanes' limbs of the SAME number: x_i*x_j is asked for at row i by
// the lane that holds limb j AND at row j by the lane that holds limb i (phe/util.py:50 gmpy2.powmod -> mpz_powm ->
// mpn_sqr_basecase computes it once).  Taking the products above the diagonal (j >= i) would idle the low lanes in lock step
// with the busy ones; instead every lane takes, in every row, the limbs of ONE residue band: with i = t*L + r and j = g*L + k
// the difference j - i is k - r modulo L whatever lane and trip, and of the two orders of a pair exactly one has
// (k - r) mod L in 1 .. ceil(L/2) - 1.  That one is taken doubled; the classes equal to their own negative (0, and L/2 for even
// L) are taken once in BOTH orders, which is the same thing (and the diagonal x_i^2 once).  L/2 + 1 multiply-adds per row
// instead of L, each row index a compile-time constant of the unrolled trip, the same work in every lane.  A column's sum at
// the row that retires it is the full product's (both rows of a pair lie at or below the pair's column); within a stay of L
// rows in one lane a column takes the weight of L products as before (its k + r is constant, so k - r runs over every class of
// one parity twice for even L, over every class once for odd L), hence the accumulator bounds of the full sweep hold.
// Measurement-only variants (tools/exp/build_variants.sh, never shipped): PHE_VARIANT_SQFULL = every product once in both orders (the
// round-4 squaring: 4 H^2); PHE_VARIANT_SQROW = the doubled digit read from a second digit row in LDS (split_square fills it)
// instead of one shift per row on the vector pipe — 18 instructions fewer per trip of 1,392 and still 1.2 % SLOWER on the same box
// (630.9 k against 638.4 k encrypts/s; full squares 586.6 k: profiles/r05c_ab_squaring.txt): the second row's stores and reads
// sit on the squaring's critical path between two sweeps, the shifts do not.
#if defined(PHE_VARIANT_SQFULL)
constexpr bool kSqSymmetric = false;
#else
constexpr bool kSqSymmetric = true;
#endif
#if defined(PHE_VARIANT_SQROW)
constexpr bool kSqDoubledRow = true;
#else
constexpr bool kSqDoubledRow = false;
#endif
template <int L>
constexpr bool sq_has_doubled() { return kSqSymmetric && L >= 3; }  // (L = 1, 2: every class is its own negative)
template <int L>
constexpr int sq_weight(int k, int r) {  // limb k of the lane, row r of the trip: 0 = the mirror image takes it, 1 = once, 2 = doubled
    if (!kSqSymmetric) return 1;
    const int c = ((k - r) % L + L) % L;
    if (c == 0 || 2 * c == L) return 1;
    return 2 * c < L ? 2 : 0;
}
// acc[(k + j) % L] += x * b[k] over the limbs a squaring row takes (x: the row's digit of X0, b: the lane's limbs of X0)
template <int L>
PHE_DEV void sq_row(uint64_t (&acc)[L], uint32_t x, uint32_t x2, const uint32_t (&b)[L], int j) {  // x2 = 2*x (limbs stay below 2^29 + 2^8)
#pragma unroll
    for (int k = 0; k < L; ++k) {
        const int w = sq_weight<L>(k, j);
        if (w == 2) acc[(k + j) % L] = wave::mad64(x2, b[k], acc[(k + j) % L]);
        else if (w == 1) acc[(k + j) % L] = wave::mad64(x, b[k], acc[(k + j) % L]);
    }
}

// ---- single-word passes (used by the conversions out of the pair form) -----------------------------------------
// out = (a*b + m*n) / R with the quotient digits m_i stored to m_row (LDS, H words); a: H digits in LDS.
// SQ: b holds the limbs of the number whose digits a holds (a squaring: sq_row)
template <int G, int L, bool U = false, bool SQ = false>
PHE_DEV void montmul_q(uint32_t (&out)[L], const uint32_t* a, const uint32_t (&b)[L], uint32_t* m_row,
                       const uint32_t (&n)[L], uint32_t n0inv, const Lanes<G>& ln, int rows = G * L) {
    uint64_t acc[L];
#pragma unroll
    for (int k = 0; k < L; ++k) acc[k] = 0;
#pragma unroll 1
    for (int i = 0; i < rows; i += L) {
        uint32_t mq[L];
#pragma unroll
        for (int j = 0; j < L; ++j) {
            const uint32_t ai = a[i + j];
            if constexpr (SQ) {
                sq_row<L>(acc, ai, ai << 1, b, j);
            } else {
#pragma unroll
                for (int k = 0; k < L; ++k) acc[(k + j) % L] = wave::mad64(ai, b[k], acc[(k + j) % L]);
            }
            const uint32_t m = wave::grp_bcast0<G>((U ? (uint32_t)acc[j] : (uint32_t)acc[j] * n0inv) & kLimbMask, ln);
            mq[j] = m;
#pragma unroll
            for (int k = 0; k < L; ++k) acc[(k + j) % L] = wave::mad64(m, n[k], acc[(k + j) % L]);
            const uint64_t low = acc[j];
            const uint32_t recv = wave::grp_down1<G>((uint32_t)low & kLimbMask, ln);
            if constexpr (L > 1) {
                acc[(j + 1) % L] += low >> kRadixBits;
                acc[j] = recv;
            } else {
                acc[0] = (low >> kRadixBits) + recv;
            }
        }
        if (ln.g == 0u) {
#pragma unroll
            for (int j = 0; j < L; ++j) m_row[i + j] = mq[j];
        }
    }
    normalize_partial<G, L>(out, acc, ln);
}

// out = (a*b + m*g + m2*n) / R;  a and m: H digits each in LDS
template <int G, int L>
PHE_DEV void montmac2(uint32_t (&out)[L], const uint32_t* a, const uint32_t (&b)[L], const uint32_t* m_row,
                      const uint32_t (&gm)[L], const uint32_t (&n)[L], uint32_t n0inv, const Lanes<G>& ln, int rows = G * L) {
    uint64_t acc[L];
#pragma unroll
    for (int k = 0; k < L; ++k) acc[k] = 0;
#pragma unroll 1
    for (int i = 0; i < rows; i += L) {
#pragma unroll
        for (int j = 0; j < L; ++j) {
            const uint32_t ai = a[i + j];
            const uint32_t mi = m_row[i + j];
#pragma unroll
            for (int k = 0; k < L; ++k) acc[(k + j) % L] = wave::mad64(ai, b[k], acc[(k + j) % L]);
#pragma unroll
            for (int k = 0; k < L; ++k) acc[(k + j) % L] = wave::mad64(mi, gm[k], acc[(k + j) % L]);
            const uint32_t m2 = wave::grp_bcast0<G>(((uint32_t)acc[j] * n0inv) & kLimbMask, ln);
#pragma unroll
            for (int k = 0; k < L; ++k) acc[(k + j) % L] = wave::mad64(m2, n[k], acc[(k + j) % L]);
            const uint64_t low = acc[j];
            const uint32_t recv = wave::grp_down1<G>((uint32_t)low & kLimbMask, ln);
            if constexpr (L > 1) {
                acc[(j + 1) % L] += low >> kRadixBits;
                acc[j] = recv;
            } else {
                acc[0] = (low >> kRadixBits) + recv;
            }
        }
    }
    normalize_partial<G, L>(out, acc, ln);
}

// out = (addend + a*b + m2*n) / R;  a and the addend: H digits each in LDS (the addend is the quotient of a previous
// montmul_q: the second word of a pair product when the sweeps are not fused)
template <int G, int L, bool U = false>
PHE_DEV void montmul_addend(uint32_t (&out)[L], const uint32_t* a, const uint32_t (&b)[L], const uint32_t* addend_row,
                            const uint32_t (&n)[L], uint32_t n0inv, const Lanes<G>& ln, int rows = G * L) {
    const uint32_t dmask = kLimbMask & ln.not_top;
    const uint32_t vmask = digit_mask<G>(ln);
    uint64_t acc[L];
#pragma unroll
    for (int k = 0; k < L; ++k) acc[k] = addend_row[ln.g * L + k];
#pragma unroll 1
    for (int i = 0; i < rows; i += L) {
#pragma unroll
        for (int j = 0; j < L; ++j) {
            const uint32_t ai = a[i + j];
#pragma unroll
            for (int k = 0; k < L; ++k) acc[(k + j) % L] = wave::mad64(ai, b[k], acc[(k + j) % L]);
            const uint32_t m2 = wave::grp_bcast0<G>(U ? (uint32_t)acc[j] : (uint32_t)acc[j] * n0inv, ln) & vmask;
#pragma unroll
            for (int k = 0; k < L; ++k) acc[(k + j) % L] = wave::mad64(m2, n[k], acc[(k + j) % L]);
            const uint64_t low = acc[j];
            const uint32_t recv = wave::grp_down1_raw<G>((uint32_t)low) & dmask;
            if constexpr (L > 1) {
                acc[(j + 1) % L] += low >> kRadixBits;
                acc[j] = recv;
            } else {
                acc[0] = (low >> kRadixBits) + recv;
            }
        }
    }
    normalize_partial<G, L>(out, acc, ln);
}

// plain product: a*b + addend = hi*R + lo.  a: H digits in LDS; lo: H canonical digits stored to lo_row (LDS; digit d at
// lo_row[d * lo_stride]: mul_tile.h keeps the digits of 64 numbers side by side); hi: almost-normalised.
template <int G, int L>
PHE_DEV void mul_wide(uint32_t (&hi)[L], const uint32_t* a, const uint32_t (&b)[L], const uint32_t (&addend)[L],
                      uint32_t* lo_row, const Lanes<G>& ln, int rows = G * L, int lo_stride = 1) {
    uint64_t acc[L];
#pragma unroll
    for (int k = 0; k < L; ++k) acc[k] = addend[k];
#pragma unroll 1
    for (int i = 0; i < rows; i += L) {
        uint32_t dq[L];
#pragma unroll
        for (int j = 0; j < L; ++j) {
            const uint32_t ai = a[i + j];
#pragma unroll
            for (int k = 0; k < L; ++k) acc[(k + j) % L] = wave::mad64(ai, b[k], acc[(k + j) % L]);
            const uint64_t low = acc[j];
            const uint32_t digit = (uint32_t)low & kLimbMask;  // final in lane 0: every lower column is done
            dq[j] = digit;
            const uint32_t recv = wave::grp_down1<G>(digit, ln);
            if constexpr (L > 1) {
                acc[(j + 1) % L] += low >> kRadixBits;
                acc[j] = recv;
            } else {
                acc[0] = (low >> kRadixBits) + recv;
            }
        }
        if (ln.g == 0u) {
#pragma unroll
            for (int j = 0; j < L; ++j) lo_row[(i + j) * lo_stride] = dq[j];
        }
    }
    normalize_partial<G, L>(hi, acc, ln);
}

// ---- pair products: both words in one sweep -------------------------------------------------------------------
// shift of one column-accumulator set by one digit (the tail of every row of a Montgomery sweep)
// (dmask = kLimbMask & ln.not_top: the digit mask and the "top lane receives 0" mask applied as one v_and)
template <int G, int L>
PHE_DEV void shift_row(uint64_t (&acc)[L], int j, uint32_t dmask) {
    const uint64_t low = acc[j];
    const uint32_t recv = wave::grp_down1_raw<G>((uint32_t)low) & dmask;
    if constexpr (L > 1) {
        acc[(j + 1) % L] += low >> kRadixBits;
        acc[j] = recv;
    } else {
        acc[0] = (low >> kRadixBits) + recv;
    }
}

// The quotient digit of the first accumulator set and its entry into the second one.  UNIT (template flag U of the sweeps):
// the modulus is the SCALED one, n' = k*n with k = -n^-1 mod 2^29, so that n' = -1 (mod 2^29) and the quotient digit is
// the accumulator's low 29 bits themselves — no v_mul_lo (key_setup.h:build_public; same-box A/B of exactly this change:
// +2.7 % encrypts/s, -2.2 % VALU instructions, profiles/r02h_ab_sweep_variants.txt).  PHE_VARIANT_QMAD: measurement-only
// variant of DESIGN 6.1 (the digit enters by one more multiply-add), built by tools/exp/build_variants.sh, never shipped.
#define PHE_SECOND_QUOTIENT(x) (U ? (uint32_t)(x) : (uint32_t)(x) * n0inv)
// lane 0's digit to every lane of the group, masked to 29 bits.  Groups inside a wave: the mask rides on the DPP move (vmask
// is VGPR data for that reason).  The whole wave (G = 64): the value passes through an SGPR anyway, so the mask is a
// scalar AND off the vector pipe — one dependent VALU step less in a chain that a lone wave cannot hide.
template <int G>
PHE_DEV uint32_t bcast_digit(uint32_t x, uint32_t vmask, const Lanes<G>& ln) {
    if constexpr (G == 64) return wave::grp_bcast0<G>(x, ln) & kLimbMask;
    else return wave::grp_bcast0<G>(x, ln) & vmask;
}
#if defined(PHE_VARIANT_QMAD)
#define PHE_QUOTIENT_STEP()                                                                  \
    const uint32_t mraw = U ? (uint32_t)p[j] : (uint32_t)p[j] * n0inv;                       \
    const uint32_t m = bcast_digit<G>(mraw, vmask, ln);                                      \
    q[j] = wave::mad64(m, wave::reread(lane0 ? 1u : 0u), q[j]);
#else
#define PHE_QUOTIENT_STEP()                                                                  \
    const uint32_t mraw = U ? (uint32_t)p[j] : (uint32_t)p[j] * n0inv;                       \
    const uint32_t m = bcast_digit<G>(mraw, vmask, ln);                                      \
    q[j] += (uint64_t)(mraw & lane0); /* quotient digit i of the first sum = digit i of the addend m */
#endif

// Digits a fused sweep takes per loop trip.  The accumulators rotate by renaming with period L, so a trip is a multiple of L
// digits: L for the throughput geometries (two waves per SIMD hide the LDS latency of the digit reads).  The whole-wave
// geometry (G = 64) runs ONE wave per SIMD and little work per digit: it takes 4-10 digits per trip and fetches the next
// trip's digits before it starts on the current ones, so that no digit read is waited for (SplitPack::rows is a multiple).
template <int G, int L>
struct Trip {
    static constexpr int kDigits = (G == 64) ? (L == 1 ? 4 : 2) * L : L;
    // fetch the next trip's digits before starting on this one's: the whole-wave geometry only.  Tried on the narrow-lane rungs
    // (L <= 9, one wave per SIMD at the batch sizes that take them) in round 3: the register copies of the hand-over cost more
    // than the LDS latency they hide — g16x5 -15 %, g16x3 -20 %, g8x9 -3 % (profiles/r03f_batch_sweep_digit_prefetch_on_small_rungs.txt)
    static constexpr bool kAhead = (G == 64);
};

// z0 = (a*b0 + m*n) / R,   z1 = (m + a*b1 + m2*n) / R.     a: H digits in LDS.
// Squaring: b0 = X0, b1 = 2*X1 (SQ: a holds the digits of b0 itself, the first word takes the symmetric half: sq_row).
// Conversion of a plain chunk a: (b0, b1) = pair(R^(j+2)).
// a2 (SQ only, may be null): the digits of a doubled, a second LDS row — a digit read instead of a shift on the vector pipe per row
template <int G, int L, bool U = false, bool SQ = false>
PHE_DEV void pair_pass2(uint32_t (&z0)[L], uint32_t (&z1)[L], const uint32_t* a, const uint32_t (&b0)[L],
                        const uint32_t (&b1)[L], const uint32_t (&n)[L], uint32_t n0inv, const Lanes<G>& ln, int rows = G * L,
                        const uint32_t* a2 = nullptr) {
    const uint32_t lane0 = kLimbMask & ~ln.not_low;  // digit mask in lane 0 of the group, 0 elsewhere
    const uint32_t dmask = kLimbMask & ln.not_top;
    // = kLimbMask in every lane (no lane of a group of >= 2 is both top and low), but plain VGPR data to the compiler,
    // so that "dpp(x) & mask" becomes one v_and_b32_dpp instead of v_and (literal) + v_mov_b32_dpp
    const uint32_t vmask = digit_mask<G>(ln);
    uint64_t p[L], q[L];
#pragma unroll
    for (int k = 0; k < L; ++k) p[k] = q[k] = 0;
    constexpr int kT = Trip<G, L>::kDigits;
    constexpr bool kSqRow2 = SQ && kSqDoubledRow && sq_has_doubled<L>() && !Trip<G, L>::kAhead;
    uint32_t ahead_a[kT];
    if constexpr (Trip<G, L>::kAhead) {
#pragma unroll
        for (int t = 0; t < kT; ++t) ahead_a[t] = a[t];
    }
#pragma unroll 1
    for (int i = 0; i < rows; i += kT) {
        uint32_t dig_a[kT];
        if constexpr (Trip<G, L>::kAhead) {
            const int nx = (i + kT < rows) ? i + kT : i;
#pragma unroll
            for (int t = 0; t < kT; ++t) {
                dig_a[t] = ahead_a[t];
                ahead_a[t] = a[nx + t];
            }
        }
#pragma unroll
        for (int jj = 0; jj < kT; ++jj) {
            const int j = jj % L;
            const uint32_t ai = Trip<G, L>::kAhead ? dig_a[jj] : a[i + jj];
            if constexpr (SQ) {
                sq_row<L>(p, ai, kSqRow2 ? a2[i + jj] : ai << 1, b0, j);  // (kSqRow2: the caller filled a2, split_square)
            } else {
#pragma unroll
                for (int k = 0; k < L; ++k) p[(k + j) % L] = wave::mad64(ai, b0[k], p[(k + j) % L]);
            }
#pragma unroll
            for (int k = 0; k < L; ++k) q[(k + j) % L] = wave::mad64(ai, b1[k], q[(k + j) % L]);
            PHE_QUOTIENT_STEP()
#pragma unroll
            for (int k = 0; k < L; ++k) p[(k + j) % L] = wave::mad64(m, n[k], p[(k + j) % L]);
            const uint32_t m2 = bcast_digit<G>(PHE_SECOND_QUOTIENT(q[j]), vmask, ln);
#pragma unroll
            for (int k = 0; k < L; ++k) q[(k + j) % L] = wave::mad64(m2, n[k], q[(k + j) % L]);
            shift_row<G, L>(p, j, dmask);
            shift_row<G, L>(q, j, dmask);
        }
    }
    normalize_partial<G, L>(z0, p, ln);
    normalize_partial<G, L>(z1, q, ln);
}

// z0 = (a*b0 + m*n) / R,   z1 = (m + a*b1 + c*b0 + m2*n) / R.     a, c: H digits each in LDS.
// Product (X0, X1) * (Y0, Y1): a = X0, c = X1, b = (Y0, Y1).
template <int G, int L, bool U = false>
PHE_DEV void pair_pass3(uint32_t (&z0)[L], uint32_t (&z1)[L], const uint32_t* a, const uint32_t* c,
                        const uint32_t (&b0)[L], const uint32_t (&b1)[L], const uint32_t (&n)[L], uint32_t n0inv,
                        const Lanes<G>& ln, int rows = G * L) {
    const uint32_t lane0 = kLimbMask & ~ln.not_low;
    const uint32_t dmask = kLimbMask & ln.not_top;
    const uint32_t vmask = digit_mask<G>(ln);
    uint64_t p[L], q[L];
#pragma unroll
    for (int k = 0; k < L; ++k) p[k] = q[k] = 0;
    constexpr int kT = Trip<G, L>::kDigits;
    uint32_t ahead_a[kT], ahead_c[kT];
    if constexpr (Trip<G, L>::kAhead) {
#pragma unroll
        for (int t = 0; t < kT; ++t) {
            ahead_a[t] = a[t];
            ahead_c[t] = c[t];
        }
    }
#pragma unroll 1
    for (int i = 0; i < rows; i += kT) {
        uint32_t dig_a[kT], dig_c[kT];
        if constexpr (Trip<G, L>::kAhead) {
            const int nx = (i + kT < rows) ? i + kT : i;
#pragma unroll
            for (int t = 0; t < kT; ++t) {
                dig_a[t] = ahead_a[t];
                dig_c[t] = ahead_c[t];
                ahead_a[t] = a[nx + t];
                ahead_c[t] = c[nx + t];
            }
        }
#pragma unroll
        for (int jj = 0; jj < kT; ++jj) {
            const int j = jj % L;
            const uint32_t ai = Trip<G, L>::kAhead ? dig_a[jj] : a[i + jj];
            const uint32_t ci = Trip<G, L>::kAhead ? dig_c[jj] : c[i + jj];
#pragma unroll
            for (int k = 0; k < L; ++k) p[(k + j) % L] = wave::mad64(ai, b0[k], p[(k + j) % L]);
#pragma unroll
            for (int k = 0; k < L; ++k) q[(k + j) % L] = wave::mad64(ai, b1[k], q[(k + j) % L]);
#pragma unroll
            for (int k = 0; k < L; ++k) q[(k + j) % L] = wave::mad64(ci, b0[k], q[(k + j) % L]);
            PHE_QUOTIENT_STEP()
#pragma unroll
            for (int k = 0; k < L; ++k) p[(k + j) % L] = wave::mad64(m, n[k], p[(k + j) % L]);
            const uint32_t m2 = bcast_digit<G>(PHE_SECOND_QUOTIENT(q[j]), vmask, ln);
#pragma unroll
            for (int k = 0; k < L; ++k) q[(k + j) % L] = wave::mad64(m2, n[k], q[(k + j) % L]);
            shift_row<G, L>(p, j, dmask);
            shift_row<G, L>(q, j, dmask);
        }
    }
    normalize_partial<G, L>(z0, p, ln);
    normalize_partial<G, L>(z1, q, ln);
}

template <int G, int L, bool U = false>
struct SplitLane {  // what every pass needs, loaded once per kernel (U: the modulus is the scaled one, see PHE_QUOTIENT_STEP)
    uint32_t n[L];
    uint32_t n0inv;
    uint32_t* row_a;  // H words: digits of X0 (or of a plain multiplier)
    uint32_t* row_c;  // H words: digits of X1 (quotient digits in split_exit)
    int rows_;        // G >= 16 only: limbs the numbers really have (R = 2^(29 rows), a multiple of L); narrower groups fill their G*L limbs
    PHE_DEV int rows() const {
        if constexpr (G >= 16) return rows_;
        else return G * L;
    }
};

// (z0, z1) = (a*b0 + m*n, m + a*b1 + m2*n) / R with a already in row_a: one fused sweep, or two single sweeps with the
// quotient digits handed over through row_c when the lanes are too wide for the fused one
template <int G, int L, bool U, bool SQ = false>
PHE_DEV void pair_mul_plain(uint32_t (&z0)[L], uint32_t (&z1)[L], const uint32_t (&b0)[L], const uint32_t (&b1)[L],
                            const SplitLane<G, L, U>& K, const Lanes<G>& ln) {
    if constexpr (L <= kMaxFusedPass2L) {
        pair_pass2<G, L, U, SQ>(z0, z1, K.row_a, b0, b1, K.n, K.n0inv, ln, K.rows(), SQ ? K.row_c : nullptr);
    } else {
        uint32_t u[L];
        montmul_q<G, L, U, SQ>(u, K.row_a, b0, K.row_c, K.n, K.n0inv, ln, K.rows());
        wave::lds_fence();
        montmul_addend<G, L, U>(z1, K.row_a, b1, K.row_c, K.n, K.n0inv, ln, K.rows());
#pragma unroll
        for (int k = 0; k < L; ++k) z0[k] = u[k];
    }
}

template <int G, int L, bool U>
PHE_DEV void split_square(uint32_t (&X0)[L], uint32_t (&X1)[L], const SplitLane<G, L, U>& K, const Lanes<G>& ln) {
    uint32_t d[L];
    if constexpr (L <= kMaxFusedL && kSqDoubledRow && sq_has_doubled<L>() && !Trip<G, L>::kAhead) {
        // digits of X0 in row_a, the same doubled in row_c (free during a fused squaring): the sweep reads both
        wave::lds_fence();
#pragma unroll
        for (int k = 0; k < L; ++k) {
            K.row_a[ln.g * L + k] = X0[k];
            K.row_c[ln.g * L + k] = X0[k] << 1;
        }
        wave::lds_fence();
    } else {
        lds_put<L>(K.row_a, X0, ln.g);
    }
#pragma unroll
    for (int k = 0; k < L; ++k) d[k] = X1[k];
    add_normalize<G, L>(d, X1, ln);  // 2*X1
    pair_mul_plain<G, L, U, true>(X0, X1, X0, d, K, ln);  // (row_a holds the very limbs of X0: the symmetric half is exact)
}

template <int G, int L, bool U>
PHE_DEV void split_mul(uint32_t (&X0)[L], uint32_t (&X1)[L], const uint32_t (&Y0)[L], const uint32_t (&Y1)[L],
                       const SplitLane<G, L, U>& K, const Lanes<G>& ln) {
    if constexpr (L <= kMaxFusedL) {
        wave::lds_fence();
#pragma unroll
        for (int k = 0; k < L; ++k) {
            K.row_a[ln.g * L + k] = X0[k];
            K.row_c[ln.g * L + k] = X1[k];
        }
        wave::lds_fence();
        pair_pass3<G, L, U>(X0, X1, K.row_a, K.row_c, Y0, Y1, K.n, K.n0inv, ln, K.rows());
    } else {
        // three single sweeps: X1*Y0, then (X0*Y0 with its quotient) and (quotient + X0*Y1); the second word stays
        // below 3n instead of 2n, which every operation accepts
        uint32_t t[L];
        lds_put<L>(K.row_a, X1, ln.g);
        montmul<G, L>(t, K.row_a, Y0, K.n, K.n0inv, ln, K.rows());
        lds_put<L>(K.row_a, X0, ln.g);
        pair_mul_plain<G, L>(X0, X1, Y0, Y1, K, ln);
        add_normalize<G, L>(X1, t, ln);
    }
}

// the number in the 32-bit-word row src -> pair representation
template <int G, int L, bool U>
PHE_DEV void split_conv(uint32_t (&X0)[L], uint32_t (&X1)[L], const uint32_t* src, int limbs32, int chunks,
                        const SplitConsts& C, const SplitLane<G, L, U>& K, const Lanes<G>& ln) {
    constexpr int H = G * L;
    uint32_t tmp[L], d0[L], d1[L], u[L], t[L];
    for (int j = 0; j < chunks; ++j) {
        load_u32_as_r29<L>(tmp, src, limbs32, j * K.rows(), ln.g, K.rows());
        lds_put<L>(K.row_a, tmp, ln.g);
        load_row<L>(d0, wave::reread_ptr(C.conv) + (size_t)(2 * j) * H, ln.g);
        load_row<L>(d1, wave::reread_ptr(C.conv) + (size_t)(2 * j + 1) * H, ln.g);
        if (j == 0) {
            pair_mul_plain<G, L>(X0, X1, d0, d1, K, ln);
        } else {
            pair_mul_plain<G, L>(u, t, d0, d1, K, ln);
            add_normalize<G, L>(X0, u, ln);
            add_normalize<G, L>(X1, t, ln);
        }
    }
    if (chunks > 1) {  // the sums exceed the lazy bounds: one product with the pair of 1 restores them
        load_row<L>(d0, wave::reread_ptr(C.e), ln.g);
        load_row<L>(d1, wave::reread_ptr(C.e) + H, ln.g);
        split_mul<G, L>(X0, X1, d0, d1, K, ln);
    }
}

// canonical 2H-limb number (lo, hi) <- (lo, hi) - (mlo, mhi) if that is not negative
template <int G, int L>
PHE_DEV void cond_sub_pair(uint32_t (&lo)[L], uint32_t (&hi)[L], const uint32_t (&mlo)[L], const uint32_t (&mhi)[L],
                           const Lanes<G>& ln) {
    uint32_t dl[L], dh[L];
    uint32_t br = 0, nz = 0;
#pragma unroll
    for (int k = 0; k < L; ++k) {
        const uint32_t v = lo[k] - mlo[k] - br;
        br = v >> 31;
        dl[k] = v & kLimbMask;
        nz |= dl[k];
    }
    uint64_t out_lo, out_hi;
    const uint64_t bin_lo = group_carry_in<G>(wave::ballot(br != 0), wave::ballot(nz == 0), out_lo);
    uint32_t bi = lane_bit(bin_lo, ln.lane);
#pragma unroll
    for (int k = 0; k < L; ++k) {
        const uint32_t v = dl[k] - bi;
        bi = v >> 31;
        dl[k] = v & kLimbMask;
    }
    // the low half's borrow enters lane 0 of the high half
    br = (ln.g == 0u) ? group_top_bit<G>(out_lo, ln.lane) : 0u;
    nz = 0;
#pragma unroll
    for (int k = 0; k < L; ++k) {
        const uint32_t v = hi[k] - mhi[k] - br;
        br = v >> 31;
        dh[k] = v & kLimbMask;
        nz |= dh[k];
    }
    const uint64_t bin_hi = group_carry_in<G>(wave::ballot(br != 0), wave::ballot(nz == 0), out_hi);
    const uint64_t take = ~out_hi & GroupMasks<G>::top;  // no final borrow: value >= modulus
    const uint32_t sel = group_top_bit<G>(take, ln.lane);
    bi = lane_bit(bin_hi, ln.lane);
#pragma unroll
    for (int k = 0; k < L; ++k) {
        const uint32_t v = dh[k] - bi;
        bi = v >> 31;
        hi[k] = sel ? (v & kLimbMask) : hi[k];
        lo[k] = sel ? dl[k] : lo[k];
    }
}

// canonical limbs (lo, hi) -> little-endian 32-bit words at p, repacked through the group's 2H-word LDS row
template <int G, int L>
PHE_DEV void store_pair_as_u32(uint32_t* p, int limbs32, const uint32_t (&lo)[L], const uint32_t (&hi)[L],
                               uint32_t* row, uint32_t g, bool live, int rows = G * L) {
    constexpr int H = G * L;
    const int S2 = 2 * rows;  // limbs of the number lo + hi * 2^(29 rows); limb q sits at row[q] (q < rows) or row[H + q - rows]
    g = wave::reread(g);
    wave::lds_fence();
#pragma unroll
    for (int k = 0; k < L; ++k) {
        row[g * L + k] = lo[k];
        row[H + g * L + k] = hi[k];
    }
    wave::lds_fence();
    if (live) {
        const auto limb = [&](int q) -> uint64_t { return q < S2 ? row[q < rows ? q : H + (q - rows)] : 0u; };
        for (int j = (int)g; j < limbs32; j += G) {
            const int bit = 32 * j;
            const int q = bit / kRadixBits, o = bit - q * kRadixBits;
            const uint64_t v = limb(q) | (limb(q + 1) << kRadixBits) | (limb(q + 2) << (2 * kRadixBits));
            p[j] = (uint32_t)(v >> o);
        }
    }
    wave::lds_fence();
}

// pair -> canonical residue of x * (1 + n*mp) mod n^2 (mp == nullptr: of x), written as 32-bit words
template <int G, int L, bool U>
PHE_DEV void split_exit(uint32_t* out, int out_limbs, uint32_t (&X0)[L], uint32_t (&X1)[L], const uint32_t* mp,
                        int mp_limbs, const SplitConsts& C, const SplitLane<G, L, U>& K, const Lanes<G>& ln, bool live,
                        const uint32_t* x1_factor = nullptr) {
    constexpr int H = G * L;
    const uint32_t g = ln.g;
    uint32_t u[L], t[L], cst[L];
    // X0 / R = u - n * m / R
    lds_put<L>(K.row_a, X0, g);
#pragma unroll
    for (int k = 0; k < L; ++k) cst[k] = (g == 0u && k == 0) ? 1u : 0u;
    montmul_q<G, L>(u, K.row_a, cst, K.row_c, K.n, K.n0inv, ln, K.rows());
    // t = -(X1 + m) / R = (X1 + m) * (n - 1) / R   (mod n)
    lds_put<L>(K.row_a, X1, g);  // (its fences also order the quotient digits in row_c)
#pragma unroll
    for (int k = 0; k < L; ++k) cst[k] = K.n[k] - ((g == 0u && k == 0) ? 1u : 0u);  // n is odd: no borrow
    if constexpr (L <= kMaxFusedL) {
        if (x1_factor != nullptr) {  // the pair came from the scaled modulus n~ = k*n: its second word is X1/k, so (n-1) -> k*(n-1)
            uint32_t kx[L];
            load_row<L>(kx, x1_factor, g);
            montmac2<G, L>(t, K.row_a, kx, K.row_c, cst, K.n, K.n0inv, ln, K.rows());
        } else {
            montmac2<G, L>(t, K.row_a, cst, K.row_c, cst, K.n, K.n0inv, ln, K.rows());
        }
    } else {
        uint32_t t2[L];
        montmul<G, L>(t, K.row_a, cst, K.n, K.n0inv, ln, K.rows());
        montmul<G, L>(t2, K.row_c, cst, K.n, K.n0inv, ln, K.rows());
        add_normalize<G, L>(t, t2, ln);
    }
    if (mp != nullptr) {  // + mp * X0 / R: the plaintext term of (1 + n*mp), phe/paillier.py:134
        uint32_t w[L];
        load_u32_as_r29<L>(w, mp, mp_limbs, 0, g, K.rows());
        lds_put<L>(K.row_a, w, g);
        montmul<G, L>(w, K.row_a, X0, K.n, K.n0inv, ln, K.rows());
        add_normalize<G, L>(t, w, ln);
    }
    // t mod n, canonical
    lds_put<L>(K.row_a, t, g);
    load_row<L>(cst, wave::reread_ptr(C.r1), g);
    montmul<G, L>(t, K.row_a, cst, K.n, K.n0inv, ln, K.rows());
    canonicalize<G, L>(t, K.n, ln);
    // v = u + n*t  (< n^2 + 2n), then the canonical residue
    lds_put<L>(K.row_a, t, g);
    uint32_t hi[L], lo[L];
    mul_wide<G, L>(hi, K.row_a, K.n, u, K.row_c, ln, K.rows());
    wave::lds_fence();
    load_row<L>(lo, K.row_c, g);
    if constexpr (G >= 16) {  // the sweep wrote `rows` digits: what lies beyond in the row is not part of the number
#pragma unroll
        for (int k = 0; k < L; ++k) lo[k] = ((int)g * L + k < K.rows()) ? lo[k] : 0u;
    }
    normalize_full<G, L>(hi, ln);
    load_row<L>(cst, wave::reread_ptr(C.nsq), g);
    load_row<L>(t, wave::reread_ptr(C.nsq) + H, g);
    cond_sub_pair<G, L>(lo, hi, cst, t, ln);
    store_pair_as_u32<G, L>(out, out_limbs, lo, hi, K.row_a, g, live, K.rows());
}

// ---- the batched exponentiation -------------------------------------------------------------------------------
// The window table of one number: entry j = the pair (X0 | X1), 2 G L words.  Limb groups keep an entry's words consecutive (lane g
// reads its L limbs of each word: the group reads 2 G L consecutive words).  ONE lane per number (G = 1) would read 64 entries of 64
// different numbers with one instruction — 64 rows 2 L words apart; there the 64 tables of a wave are interleaved,
// [entry][word][lane]: every read and write of the wave is 256 consecutive bytes.
template <int G, int L>
struct WindowTable {
    static constexpr int H = G * L, S2 = 2 * H;
    uint32_t* base;
    PHE_DEV WindowTable(uint32_t* table, uint32_t slot, uint32_t lane, int entries) {
        if constexpr (G == 1) base = table + (size_t)(slot - lane) * (size_t)entries * S2 + lane;  // (a wave's slots are consecutive)
        else base = table + (size_t)slot * (size_t)entries * S2;
    }
    PHE_DEV void store(int j, const uint32_t (&x0)[L], const uint32_t (&x1)[L], uint32_t g) const {
        if constexpr (G == 1) {
#pragma unroll
            for (int k = 0; k < L; ++k) {
                base[((size_t)j * S2 + k) * 64] = x0[k];
                base[((size_t)j * S2 + H + k) * 64] = x1[k];
            }
        } else {
            store_row<L>(base + (size_t)j * S2, x0, g);
            store_row<L>(base + (size_t)j * S2 + H, x1, g);
        }
    }
    PHE_DEV void load(uint32_t (&x0)[L], uint32_t (&x1)[L], int j, uint32_t g) const {
        if constexpr (G == 1) {
#pragma unroll
            for (int k = 0; k < L; ++k) {
                x0[k] = base[((size_t)j * S2 + k) * 64];
                x1[k] = base[((size_t)j * S2 + H + k) * 64];
            }
        } else {
            load_row<L>(x0, base + (size_t)j * S2, g);
            load_row<L>(x1, base + (size_t)j * S2 + H, g);
        }
    }
};

template <int G, int L, int MODE, bool U = false>
PHE_DEV void modexp_split_body(const SplitArgs& A, uint32_t* lds_row, uint32_t slot, uint32_t total_slots,
                               uint32_t lane) {
    constexpr int H = G * L;
    const Lanes<G> ln(lane);
    const uint32_t g = ln.g;
    SplitLane<G, L, U> K;
    load_row<L>(K.n, A.mod.n, g);
    K.n0inv = A.mod.n0inv;
    K.rows_ = A.mod.rows;
    K.row_a = lds_row;
    K.row_c = lds_row + H;
    const WindowTable<G, L> tbl(A.table, slot, lane, A.tbl_entries);
    const uint64_t n_iter = (A.batch + total_slots - 1) / total_slots;
    for (uint64_t it = 0; it < n_iter; ++it) {
        uint64_t item = slot + it * (uint64_t)total_slots;
        const bool live = item < A.batch;
        if (!live) item = A.batch - 1;
        uint32_t X0[L], X1[L], Y0[L], Y1[L];
        split_conv<G, L>(X0, X1, A.base + item * (uint64_t)A.base_limbs, A.base_limbs, A.base_chunks, A.mod, K, ln);
        // ---- odd powers base^1, base^3, ... ---------------------------------------------------------------
        tbl.store(0, X0, X1, g);
        if (A.tbl_entries > 1) {
#pragma unroll
            for (int k = 0; k < L; ++k) {
                Y0[k] = X0[k];
                Y1[k] = X1[k];
            }
            split_square<G, L>(Y0, Y1, K, ln);  // base^2
            for (int j = 1; j < A.tbl_entries; ++j) {
                split_mul<G, L>(X0, X1, Y0, Y1, K, ln);
                tbl.store(j, X0, X1, g);
            }
        }
        // ---- left-to-right sliding window ------------------------------------------------------------------
        tbl.load(X0, X1, A.first_idx, g);
        for (int op = 0; op < A.n_ops; ++op) {
            const uint32_t w = A.sched[op];
            const int nsq = (int)(w >> 8);
            const int sel = (int)(w & 0xffu);
            for (int s = 0; s < nsq; ++s) split_square<G, L>(X0, X1, K, ln);
            if (sel) {
                tbl.load(Y0, Y1, sel - 1, g);
                split_mul<G, L>(X0, X1, Y0, Y1, K, ln);
            }
        }
        // ---- the op's final factor and the way out of the pair representation --------------------------------
        const uint32_t* mp = nullptr;
        if (MODE == kModeEncrypt) {
            // nude ciphertext 1 + n*m folded into split_exit (no plaintexts given: the bare power base^e)
            mp = A.post ? A.post + item * (uint64_t)A.post_limbs : nullptr;
        } else if (MODE == kModeObfuscate) {
            split_conv<G, L>(Y0, Y1, A.post + item * (uint64_t)A.post_limbs, A.post_limbs, A.post_chunks, A.mod, K, ln);
            split_mul<G, L>(X0, X1, Y0, Y1, K, ln);
        }
        bool quick = false;
        PHE_BOUNDS(G >= 16 || A.exit_mod.n == nullptr);  // (phe_hip.hip prepare_late refuses narrower rungs: they hold no quick way out)
        if constexpr (U && G >= 16) quick = A.exit_mod.n != nullptr;  // (only 16-lane rungs are launched with it: phe_hip.hip prepare_late;
                                                                      //  a compile-time no for the narrow groups keeps the second copy of
                                                                      //  the way out — ~5 k instructions — out of the throughput kernels)
        if (quick) {
            // "quick" rungs (16-lane groups on the scaled modulus with R = 2^(29 rows), rows = the limbs it needs: key_setup.h
            // QuickPack): the way out works modulo the TRUE modulus with the same R — X0 - n~*X1 = X0 - n*(k*X1), one constant row
            // k*(n-1) where split_exit uses n-1 otherwise (see modexp_split_ab_body) — straight to the canonical residue, the
            // plaintext factor folded in: no wide scratch row, no product pass
            SplitLane<G, L, false> KE;
            load_row<L>(KE.n, A.exit_mod.n, g);
            KE.n0inv = A.exit_mod.n0inv;
            KE.rows_ = K.rows_;
            KE.row_a = K.row_a;
            KE.row_c = K.row_c;
            split_exit<G, L>(A.out + item * (uint64_t)A.out_limbs, A.out_limbs, X0, X1, mp, A.post_limbs, A.exit_mod, KE, ln, live,
                             A.exit_mod.kx);
        } else {
            split_exit<G, L>(A.out + item * (uint64_t)A.out_limbs, A.out_limbs, X0, X1, mp, A.post_limbs, A.mod, K, ln, live);
        }
    }
}

// ---- the wave pair's row order on ONE wavefront: both words of a pair product in one sweep ("late" sweeps) -------------------
// The rungs that serve small batches (groups of 16 lanes, the whole wave) run one wave per SIMD or little more, where nothing
// hides a dependent chain, and their numbers have few limbs per lane, where every instruction that is not a multiply-add
// counts.  Round 3 gave the wave-pair kernels (below) a row order for exactly that situation; these sweeps bring it to the
// single-wave kernels: the modulus is the SCALED one n~ = k*n = -1 (mod 2^29) (key_setup.h QuickPack, R = 2^(29 rows) with rows
// = the limbs n~ really needs, a multiple of L — not the G*L the lanes could hold), so a quotient digit is the accumulator's
// low digit as it is, and its product enters AFTER the shift:
//      q_s     = S_s mod 2^29
//      S_(s+1) = (S_s - q_s) / 2^29  +  a_s * b  +  q_s * nbar              nbar = (n~ + 1) / 2^29,   s = 0 ... rows, S_0 = 0
// i.e. S * R = a*b + Q*n~ with Q = sum q_(s+1) 2^(29 s): the digit travels to the other lanes while they shift and take a_s*b
// in, and ONE multiply-add per limb closes the row.  The second word adds digit s of Q at ITS step s — the digit the first word
// forms at step s + 1 — so the fused sweep runs the first word one step ahead: iteration i = first word's step i + 1 and second
// word's step i, a prologue (the first word's step 0: a_0*b0 alone) and an epilogue (the second word's step `rows`: no digit).
// Same algebra as ab_first_steps / ab_second_steps below (whose tests pin it), the same values at the end of the sweep.
template <int G, int L>
struct LateShape {
    // iterations per trip: a multiple of L (the accumulators rotate by renaming with period L)
    static constexpr int kIter = (G == 64 ? (L == 1 ? 4 : (L == 2 ? 2 : 1)) : 1) * L;
};

// products a column takes per step -> may its carry be taken as a 32-bit number (at most 7 products of < 2^58 between two shifts)?
template <int G, int L>
constexpr bool late_narrow(int products_per_step) {
    return products_per_step * L <= 7 && (L == 1 || G == 64);
}

// masks of a late sweep, plain VGPR data (see pair_pass2)
template <int G>
struct LateMasks {
    uint32_t lane0, dmask, vmask;
    PHE_DEV explicit LateMasks(const Lanes<G>& ln) {
        lane0 = kLimbMask & ~ln.not_low;
        dmask = kLimbMask & ln.not_top;
        vmask = digit_mask<G>(ln);
    }
};

// One quotient step of a column-accumulator set: the low digit of the lowest column (index jl) is, in lane 0, the quotient
// digit — broadcast as the return value —, the set moves down by one column, `extra` (lane 0: a digit of the other word's
// quotient; 0 elsewhere) joins the carry.  lane0_digit: the low digit in lane 0, 0 in the other lanes (WANT_DIGIT only).
// NARROW: the column holds at most 7 products between two shifts (< 2^61): its carry is a 32-bit number (ab_shift_narrow).
template <int G, int L, bool NARROW, bool WANT_DIGIT>
PHE_DEV uint32_t late_quotient_step(uint64_t (&acc)[L], int jl, uint32_t extra, uint32_t& lane0_digit, const LateMasks<G>& mk,
                                    const Lanes<G>& ln) {
    uint32_t m, recv;
    if constexpr (G == 64) {
        const uint32_t t = wave::reread((uint32_t)acc[jl] & kLimbMask);  // (kept on the vector side: the lane shift takes it there)
        m = wave::grp_bcast0<G>(t, ln);  // through an SGPR: no mask to apply afterwards
        recv = wave::grp_down1<G>(t, ln);
        if constexpr (WANT_DIGIT) lane0_digit = t & mk.lane0;
    } else {
        const uint32_t low = (uint32_t)acc[jl];
        m = wave::grp_bcast0<G>(low, ln) & mk.vmask;          // one v_and_b32_dpp each
        recv = wave::grp_down1_raw<G>(low) & mk.dmask;
        if constexpr (WANT_DIGIT) lane0_digit = low & mk.lane0;
    }
    if constexpr (NARROW) {
        const uint32_t carry = (uint32_t)(acc[jl] >> kRadixBits);
        if constexpr (L > 1) {
            acc[(jl + 1) % L] += (uint64_t)(carry + extra);
            acc[jl] = (uint64_t)recv;
        } else {
            acc[0] = (uint64_t)(recv + carry + extra);
        }
    } else {
        const uint64_t carry = acc[jl] >> kRadixBits;
        if constexpr (L > 1) {
            acc[(jl + 1) % L] += carry + (uint64_t)extra;
            acc[jl] = (uint64_t)recv;
        } else {
            acc[0] = carry + (uint64_t)recv + (uint64_t)extra;
        }
    }
    return m;
}

// iteration T (mod L) of a trip: the first word's step i + 1 (digit an = a_(i+1)) and the second word's step i (ai = a_i, ci = c_i)
// SQ (with MUL false): a squaring — b0 holds the limbs of the number whose digits the sweep reads, the first word takes the symmetric
// half (sq_row; the first word's step s adds a_s*b0 into the frame whose lowest column is s mod L)
template <int G, int L, bool MUL, int T, bool SQ = false>
PHE_DEV void late_iteration(uint64_t (&p)[L], uint64_t (&q)[L], uint32_t an, uint32_t ai, uint32_t ci, const uint32_t (&b0)[L],
                            const uint32_t (&b1)[L], const uint32_t (&nbar)[L], const LateMasks<G>& mk, const Lanes<G>& ln) {
    constexpr int jl = T % L, j = (T + 1) % L;         // first word: lowest column before / after the shift of its step
    constexpr int jlq = (T + L - 1) % L, jq = T % L;   // second word
    // narrow (32-bit) carries: one limb per lane always (the whole shift is 32-bit arithmetic); with more limbs only on the whole
    // wave, where a lone wavefront waits for the 64-bit shift's latency — on 16-lane groups the zero-extensions the narrow form
    // needs for its 64-bit additions cost 3-4 more instructions per row than they save (code object, round 4)
    constexpr bool kNarrowP = late_narrow<G, L>(2), kNarrowQ = late_narrow<G, L>(MUL ? 3 : 2);
    uint32_t dq = 0, unused = 0;
    const uint32_t m = late_quotient_step<G, L, kNarrowP, true>(p, jl, 0u, dq, mk, ln);    // dq: digit i of Q (lane 0)
    const uint32_t m2 = late_quotient_step<G, L, kNarrowQ, false>(q, jlq, dq, unused, mk, ln);
    if constexpr (SQ) {
        sq_row<L>(p, an, an << 1, b0, j);
    } else {
#pragma unroll
        for (int k = 0; k < L; ++k) p[(k + j) % L] = wave::mad64(an, b0[k], p[(k + j) % L]);
    }
#pragma unroll
    for (int k = 0; k < L; ++k) q[(k + jq) % L] = wave::mad64(ai, b1[k], q[(k + jq) % L]);
    if constexpr (MUL) {
#pragma unroll
        for (int k = 0; k < L; ++k) q[(k + jq) % L] = wave::mad64(ci, b0[k], q[(k + jq) % L]);
    }
    if constexpr (L == 1) {  // (the digits' products first: the quotient digits are still on their way)
        p[0] = wave::reread64(p[0]);
        q[0] = wave::reread64(q[0]);
    }
#pragma unroll
    for (int k = 0; k < L; ++k) p[(k + j) % L] = wave::mad64(m, nbar[k], p[(k + j) % L]);
#pragma unroll
    for (int k = 0; k < L; ++k) q[(k + jq) % L] = wave::mad64(m2, nbar[k], q[(k + jq) % L]);
}

template <int G, int L, bool MUL, int N, bool SQ = false, int T = 0>
PHE_DEV void late_trip(uint64_t (&p)[L], uint64_t (&q)[L], const uint32_t (&da)[N + 1], const uint32_t (&dc)[N], const uint32_t (&b0)[L],
                       const uint32_t (&b1)[L], const uint32_t (&nbar)[L], const LateMasks<G>& mk, const Lanes<G>& ln) {
    if constexpr (T < N) {
        late_iteration<G, L, MUL, T % L, SQ>(p, q, da[T + 1], da[T], dc[T], b0, b1, nbar, mk, ln);
        late_trip<G, L, MUL, N, SQ, T + 1>(p, q, da, dc, b0, b1, nbar, mk, ln);
    }
}

// z0 = (a*b0 + Q*n~) / R,   z1 = (Q + a*b1 [+ c*b0] + Q2*n~) / R;   a (and c): digit rows in LDS, `rows` digits each (a multiple of L);
// word `rows` of a must be 0 (see `load` below).   MUL: the product (a, c) * (b0, b1); else a squaring / a conversion.
template <int G, int L, bool MUL, bool SQ = false>
PHE_DEV void pair_late(uint32_t (&z0)[L], uint32_t (&z1)[L], const uint32_t* a, const uint32_t* c, const uint32_t (&b0)[L],
                       const uint32_t (&b1)[L], const uint32_t (&nbar)[L], const Lanes<G>& ln, int rows) {
    constexpr int kI = LateShape<G, L>::kIter;
    const LateMasks<G> mk(ln);
    uint64_t p[L], q[L];
#pragma unroll
    for (int k = 0; k < L; ++k) p[k] = q[k] = 0;
    // digits a_i .. a_(i+N) and c_i .. c_(i+N-1) of a trip of N iterations starting at i.  a_rows, the digit the first word's last
    // step "takes", must read as 0: it does — the word after the number's last limb is a zero limb where the lanes hold more
    // than `rows` limbs, and the row's zeroed pad word where they do not (LateLane: the caller keeps a[G*L] zero)
    const auto load = [&](auto& da, auto& dc, int i, auto n_tag) {
        constexpr int N = decltype(n_tag)::value;
#pragma unroll
        for (int t = 0; t <= N; ++t) da[t] = a[i + t];
#pragma unroll
        for (int t = 0; t < N; ++t) dc[t] = MUL ? c[i + t] : 0u;
    };
    using TagI = std::integral_constant<int, kI>;
    using TagL = std::integral_constant<int, L>;
    PHE_BOUNDS(rows >= L && rows % L == 0 && rows <= G * L);  // a_rows is the last word `load` touches: inside the row or its pad
    // prologue: the first word's step 0
    {
        const uint32_t a0 = a[0];
        if constexpr (SQ) {
            sq_row<L>(p, a0, a0 << 1, b0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < L; ++k) p[k] = wave::mad64(a0, b0[k], p[k]);
        }
    }
    int i = 0;
    if constexpr (G == 64) {
        // a lone wave per SIMD hides nothing: two digit sets that change roles, each fetched a trip ahead of its use (no copies)
        uint32_t da_a[kI + 1], dc_a[kI], da_b[kI + 1], dc_b[kI];
        if (rows >= kI) load(da_a, dc_a, 0, TagI());
#pragma unroll 1
        for (; i + 2 * kI <= rows; i += 2 * kI) {
            load(da_b, dc_b, i + kI, TagI());
            late_trip<G, L, MUL, kI, SQ>(p, q, da_a, dc_a, b0, b1, nbar, mk, ln);
            if (i + 3 * kI <= rows) load(da_a, dc_a, i + 2 * kI, TagI());
            late_trip<G, L, MUL, kI, SQ>(p, q, da_b, dc_b, b0, b1, nbar, mk, ln);
        }
        if (i + kI <= rows) {  // (set a holds this trip: fetched before the loop, or by its last pass)
            late_trip<G, L, MUL, kI, SQ>(p, q, da_a, dc_a, b0, b1, nbar, mk, ln);
            i += kI;
        }
    } else {
#pragma unroll 1
        for (; i + kI <= rows; i += kI) {
            uint32_t da[kI + 1], dc[kI];
            load(da, dc, i, TagI());
            late_trip<G, L, MUL, kI, SQ>(p, q, da, dc, b0, b1, nbar, mk, ln);
        }
    }
#pragma unroll 1
    for (; i < rows; i += L) {  // what whole trips did not cover: L iterations at a time
        uint32_t da[L + 1], dc[L];
        load(da, dc, i, TagL());
        late_trip<G, L, MUL, L, SQ>(p, q, da, dc, b0, b1, nbar, mk, ln);
    }
    // epilogue: the second word's step `rows` (no digit, and Q has no digit `rows`)
    {
        constexpr bool kNarrowQ = late_narrow<G, L>(MUL ? 3 : 2);
        uint32_t unused = 0;
        const uint32_t m2 = late_quotient_step<G, L, kNarrowQ, false>(q, L - 1, 0u, unused, mk, ln);
#pragma unroll
        for (int k = 0; k < L; ++k) q[k] = wave::mad64(m2, nbar[k], q[k]);
    }
    normalize_partial<G, L>(z0, p, ln);
    normalize_partial<G, L>(z1, q, ln);
}

// what the late sweeps need, loaded once per kernel.  LDS of a limb group (2*G*L + kLdsPad words): row_a = [0, H), then the pad
// words, then row_c = [H + kLdsPad, 2H + kLdsPad) — so that word H of row_a is a pad word that late_zero_pad keeps at zero
// (the digit "a_rows" of a sweep whose numbers fill all H limbs of the lanes).
template <int G, int L>
struct LateLane {
    uint32_t nbar[L];
    uint32_t* row_a;
    uint32_t* row_c;
    int rows;
};
template <int G, int L>
PHE_DEV void late_zero_pad(const LateLane<G, L>& K, const Lanes<G>& ln) {
    if (ln.g == 0u) K.row_a[G * L] = 0u;  // (ordered before the sweeps' reads by the fences of the next lds_put)
}

template <int G, int L>
PHE_DEV void late_square(uint32_t (&X0)[L], uint32_t (&X1)[L], const LateLane<G, L>& K, const Lanes<G>& ln) {
    uint32_t d[L];
    lds_put<L>(K.row_a, X0, ln.g);
#pragma unroll
    for (int k = 0; k < L; ++k) d[k] = X1[k];
    add_normalize<G, L>(d, X1, ln);  // 2*X1
    pair_late<G, L, false, true>(X0, X1, K.row_a, nullptr, X0, d, K.nbar, ln, K.rows);
}

template <int G, int L>
PHE_DEV void late_mul(uint32_t (&X0)[L], uint32_t (&X1)[L], const uint32_t (&Y0)[L], const uint32_t (&Y1)[L], const LateLane<G, L>& K,
                      const Lanes<G>& ln) {
    wave::lds_fence();
#pragma unroll
    for (int k = 0; k < L; ++k) {
        K.row_a[ln.g * L + k] = X0[k];
        K.row_c[ln.g * L + k] = X1[k];
    }
    wave::lds_fence();
    pair_late<G, L, true>(X0, X1, K.row_a, K.row_c, Y0, Y1, K.nbar, ln, K.rows);
}

// the number in the 32-bit-word row src -> the pair form modulo the scaled modulus (split_conv on the late sweeps)
template <int G, int L>
PHE_DEV void late_conv(uint32_t (&X0)[L], uint32_t (&X1)[L], const uint32_t* src, int limbs32, int chunks, const SplitConsts& C,
                       const LateLane<G, L>& K, const Lanes<G>& ln) {
    constexpr int H = G * L;
    uint32_t tmp[L], d0[L], d1[L], u[L], t[L];
    for (int j = 0; j < chunks; ++j) {
        load_u32_as_r29<L>(tmp, src, limbs32, j * K.rows, ln.g, K.rows);
        lds_put<L>(K.row_a, tmp, ln.g);
        load_row<L>(d0, wave::reread_ptr(C.conv) + (size_t)(2 * j) * H, ln.g);
        load_row<L>(d1, wave::reread_ptr(C.conv) + (size_t)(2 * j + 1) * H, ln.g);
        if (j == 0) {
            pair_late<G, L, false>(X0, X1, K.row_a, nullptr, d0, d1, K.nbar, ln, K.rows);
        } else {
            pair_late<G, L, false>(u, t, K.row_a, nullptr, d0, d1, K.nbar, ln, K.rows);
            add_normalize<G, L>(X0, u, ln);
            add_normalize<G, L>(X1, t, ln);
        }
    }
    if (chunks > 1) {  // the sums exceed the lazy bounds: one product with the pair of 1 restores them
        load_row<L>(d0, wave::reread_ptr(C.e), ln.g);
        load_row<L>(d1, wave::reread_ptr(C.e) + H, ln.g);
        late_mul<G, L>(X0, X1, d0, d1, K, ln);
    }
}

// modexp_split_body on the late sweeps: A.mod = constants modulo the scaled modulus n~ (with nbar), A.exit_mod = modulo the true n
// with the same R (with kx): the way out is the ordinary split_exit modulo n, fed X0 - n~*X1 = X0 - n*(k*X1) (see
// modexp_split_ab_body).  MODE: kModeEncrypt (A.post: plaintexts or nullptr for the bare power) or kModeHalfDecrypt.
template <int G, int L, int MODE>
PHE_DEV void modexp_split_late_body(const SplitArgs& A, uint32_t* lds_row, uint32_t slot, uint32_t total_slots, uint32_t lane) {
    constexpr int H = G * L, S2 = 2 * H;
    const Lanes<G> ln(lane);
    const uint32_t g = ln.g;
    LateLane<G, L> K;
    load_row<L>(K.nbar, A.mod.nbar, g);
    K.rows = A.mod.rows;
    K.row_a = lds_row;
    K.row_c = lds_row + H + kLdsPad;
    uint32_t* tbl = A.table + (size_t)slot * (size_t)A.tbl_entries * S2;
    const uint64_t n_iter = (A.batch + total_slots - 1) / total_slots;
    for (uint64_t it = 0; it < n_iter; ++it) {
        uint64_t item = slot + it * (uint64_t)total_slots;
        const bool live = item < A.batch;
        if (!live) item = A.batch - 1;
        uint32_t X0[L], X1[L], Y0[L], Y1[L];
        late_zero_pad<G, L>(K, ln);  // (the way out of the previous item wrote across the pad words)
        late_conv<G, L>(X0, X1, A.base + item * (uint64_t)A.base_limbs, A.base_limbs, A.base_chunks, A.mod, K, ln);
        // ---- odd powers base^1, base^3, ... ---------------------------------------------------------------
        store_row<L>(tbl, X0, g);
        store_row<L>(tbl + H, X1, g);
        if (A.tbl_entries > 1) {
#pragma unroll
            for (int k = 0; k < L; ++k) {
                Y0[k] = X0[k];
                Y1[k] = X1[k];
            }
            late_square<G, L>(Y0, Y1, K, ln);  // base^2
            for (int j = 1; j < A.tbl_entries; ++j) {
                late_mul<G, L>(X0, X1, Y0, Y1, K, ln);
                store_row<L>(tbl + (size_t)j * S2, X0, g);
                store_row<L>(tbl + (size_t)j * S2 + H, X1, g);
            }
        }
        // ---- left-to-right sliding window ------------------------------------------------------------------
        load_row<L>(X0, tbl + (size_t)A.first_idx * S2, g);
        load_row<L>(X1, tbl + (size_t)A.first_idx * S2 + H, g);
        for (int op = 0; op < A.n_ops; ++op) {
            const uint32_t w = A.sched[op];
            const int nsq = (int)(w >> 8);
            const int sel = (int)(w & 0xffu);
            if (sel) {  // the factor is fetched before the squarings: it arrives under them
                load_row<L>(Y0, tbl + (size_t)(sel - 1) * S2, g);
                load_row<L>(Y1, tbl + (size_t)(sel - 1) * S2 + H, g);
            }
            for (int s = 0; s < nsq; ++s) late_square<G, L>(X0, X1, K, ln);
            if (sel) late_mul<G, L>(X0, X1, Y0, Y1, K, ln);
        }
        // ---- the way out: modulo the true n -------------------------------------------------------------------------
        SplitLane<G, L, false> KE;
        load_row<L>(KE.n, A.exit_mod.n, g);
        KE.n0inv = A.exit_mod.n0inv;
        KE.rows_ = K.rows;
        KE.row_a = lds_row;      // (the way out uses the group's block as two adjacent rows of H words: store_pair_as_u32)
        KE.row_c = lds_row + H;
        const uint32_t* mp = nullptr;
        if (MODE == kModeEncrypt) mp = A.post ? A.post + item * (uint64_t)A.post_limbs : nullptr;
        split_exit<G, L>(A.out + item * (uint64_t)A.out_limbs, A.out_limbs, X0, X1, mp, A.post_limbs, A.exit_mod, KE, ln, live,
                         A.exit_mod.kx);
    }
}

// ---- one number on TWO wavefronts (the lowest-latency form, for a handful of numbers) ---------------------------------
// A pair product is two coupled chains per digit: the first word's (p += a*b0, quotient digit m, p += m*n, shift) and the
// second word's (q += a*b1 [+ c*b0] + m, its own quotient digit m2, q += m2*n, shift).  On the whole-wave geometry a wave
// issues ~18 vector instructions per digit for both and is issue-bound even alone on its SIMD ("whole-wave
// groups").  But the FIRST word never looks at the second: x~ = X0 - n*X1, and X0 of a product is the plain half-width
// Montgomery product of the X0's.  So wave A runs the first words of all products of an exponentiation — an ordinary
// Montgomery ladder modulo n — and leaves, per product, the multiplier's digits and its quotient digits in LDS; wave B, one
// product behind, runs the second words from those.  Two LDS slots and ONE workgroup barrier per product keep them in step
// (A fills slot k&1 and arrives; B arrives and reads it while A fills the other).  The conversion in and the way out need
// both words: B hands its word over in LDS.  Only for G = 64 (one number per wave pair) and fused-sweep widths.
//
// A lone wave issues in order and hides nothing, so what counts here is the dependent chain per digit.  The textbook row
// (acc += a_i*b; m = acc0 * (-n^-1); acc += m*n; shift) is ONE chain: multiply-add -> v_mul_lo -> lane 0 to an SGPR ->
// multiply-add -> lane shift -> next row, ~118 cycles per digit measured (profiles/r03d_latency_*).  These sweeps run modulo
// the SCALED modulus n~ = k*n, k = -n^-1 mod 2^29 (key_setup.h QuickPack: n~ = -1 mod 2^29, the pair form is the one
// modulo n~^2), in the order
//      q_i     = S_i mod 2^29                                   (the low digit as it is: no multiply)
//      S_(i+1) = (S_i - q_i) / 2^29  +  q_i * nbar  +  a_i * b  (nbar = (n~ + 1) / 2^29;  S_i + q_i*n~ = S_i - q_i + q_i*2^29*nbar)
// i.e. the quotient's product enters AFTER the shift: lane 0's digit travels to its SGPR while the lanes shift and take a_i*b
// in, and one multiply-add closes the row.  With S_0 = 0 (q_0 = 0) and a_rows = 0 this is rows quotient steps and rows
// products a_i*b, and S*R = a*b + Q*n~ for Q = sum q_(i+1) 2^(29 i) exactly as in the textbook order — the second word takes
// those digits of Q as its addend.  The way out (split_exit) works modulo the true n with the same R: X0 - n~*X1 = X0 - n*(k*X1).
#ifndef PHE_AB_GENERIC_STEPS
#define PHE_AB_GENERIC_STEPS 0  // 1 (tools/latency_probe.hip only): the one-limb-per-lane sweeps take the general step as well
#endif
// Step s = 0 ... rows of a sweep:   s > 0: q = S mod 2^29, S <- (S - q) / 2^29;   s < rows: S += a_s * b (+ ...);   s > 0: S += q * nbar.
// Steps come in trips of kSteps (digit s is word s of its LDS row: a trip reads ONE aligned block of every digit row); the
// first trip's step 0 has no quotient part (FIRST), the last trip has (rows + 1) - kSteps * (trips - 1) steps of which the
// last has no digit (LAST).  The quotient digit q of step s is digit s - 1 of Q, word s - 1 of the quotient row: a block of that
// row is the last kSteps - 1 digits of one trip and the first of the next (the one-limb sweeps keep them in registers and store
// the block with one aligned 16-byte store); the second word adds digit s of Q at its step s.
template <int L>
struct AbTrip {
    static constexpr int kSteps = (L == 1 ? 4 : 2) * L;
};
// LDS of a wave pair, in words: two slots of (a-digits | quotient digits), B's own digit row, one row for words handed over,
// and the area the lanes other than lane 0 write their (meaningless) copies of the quotient digits to, so that the store of
// a trip's quotient digits is an ordinary all-lane store instead of a branch around a one-lane store
// ... and H + 48 zero words: what the lanes above lane 0 read where lane 0 reads the first word's quotient digits (the digit
// blocks are fetched up to two trips ahead of the last digit: rows + 3 trips <= H + 30 words are touched)
template <int L>
constexpr int ab_lds_words() { return 6 * 64 * L + (4 * 64 + 64 * L + 16) + (64 * L + 48); }

template <int N, class P>
PHE_DEV void ab_load_block(uint32_t (&d)[N], P p) {
#pragma unroll
    for (int t = 0; t < N; ++t) d[t] = p[t];
}

// one shift of a column-accumulator set whose columns stay below 2^61 (at most 7 products of < 2^58 each between two shifts):
// the carry low >> 29 is a 32-bit number — one v_alignbit and a 32-bit-into-64-bit add instead of a 64-bit shift and add.
// t = low & mask (already formed by the caller, who also broadcasts it)
template <int L>
PHE_DEV void ab_shift_narrow(uint64_t (&acc)[L], int j, uint32_t t, uint32_t extra) {
    const uint32_t carry = (uint32_t)(acc[j] >> kRadixBits);
    const uint32_t recv = wave::grp_down1_raw<64>(t);  // (lane 63 receives 0)
    if constexpr (L > 1) {
        acc[(j + 1) % L] += (uint64_t)(carry + extra);
        acc[j] = (uint64_t)recv;
    } else {
        acc[0] = (uint64_t)(recv + carry + extra);
    }
}

// one trip of the first word: N steps on the digit block dig_a; w: this trip's block of the quotient row (step jj's digit is
// w[jj - 1]); held: the previous trip's last kSteps - 1 quotient digits, not stored yet
template <int L, int N, bool FIRST, bool LAST>
PHE_DEV void ab_first_steps(uint64_t (&p)[L], const uint32_t (&dig_a)[AbTrip<L>::kSteps], wave::lds_u32* w,
                            uint32_t (&held)[AbTrip<L>::kSteps - 1], const uint32_t (&b0)[L], const uint32_t (&nbar)[L], uint32_t dmask,
                            const Lanes<64>& ln) {
    constexpr int G = 64, kT = AbTrip<L>::kSteps;
    constexpr bool kNarrow = 2 * L <= 7 && !PHE_AB_GENERIC_STEPS;  // two products per column and step
    uint32_t tq[kT];
#pragma unroll
    for (int jj = 0; jj < kT; ++jj) tq[jj] = 0u;
#pragma unroll
    for (int jj = 0; jj < N; ++jj) {
        const bool quot = !(FIRST && jj == 0), digit = !(LAST && jj == N - 1);
        const int j = jj % L, jl = (jj + L - 1) % L;  // lowest column after / before this step's shift
        uint32_t m = 0;
        if (quot) {
            if constexpr (kNarrow) {
                const uint32_t t = wave::reread((uint32_t)p[jl] & kLimbMask);  // (kept on the vector side: the lane shift takes it there)
                m = wave::grp_bcast0<G>(t, ln);
                tq[jj] = t;  // lane 0's copy is the quotient digit
                ab_shift_narrow<L>(p, jl, t, 0u);
            } else {
                m = bcast_digit<G>((uint32_t)p[jl], 0u, ln);
                tq[jj] = m;
                shift_row<G, L>(p, jl, dmask);
            }
            if (jj == 0) {  // the previous block of the quotient row is complete: one 16-byte store, or 8-byte ones (kSteps * 4 bytes apart)
                if constexpr (kT == 4) {
                    wave::lds_store4(w - 4, held[0], held[1], held[2], tq[0]);
                } else {
#pragma unroll
                    for (int i = 0; i + 2 < kT; i += 2) wave::lds_store2(w - kT + i, held[i], held[i + 1]);
                    wave::lds_store2(w - 2, held[kT - 2], tq[0]);
                }
            }
        }
        if (digit) {
#pragma unroll
            for (int k = 0; k < L; ++k) p[(k + j) % L] = wave::mad64(dig_a[jj], b0[k], p[(k + j) % L]);
            if constexpr (L == 1) p[0] = wave::reread64(p[0]);  // (the digit's product first: the quotient digit is still on its way)
        }
        if (quot) {
#pragma unroll
            for (int k = 0; k < L; ++k) p[(k + j) % L] = wave::mad64(m, nbar[k], p[(k + j) % L]);
        }
    }
    if constexpr (LAST) {
#pragma unroll
        for (int jj = 1; jj < N; ++jj) w[jj - 1] = tq[jj];
    } else {
#pragma unroll
        for (int jj = 1; jj < kT; ++jj) held[jj - 1] = tq[jj];
    }
}

// z0 = (a*b0 + Q*n~) / R; the digits of Q go to m_row (H words of LDS; `dump`: 4*64 + H + 16 words for the other lanes' copies)
template <int L>
PHE_DEV void ab_first_word(uint32_t (&z0)[L], const uint32_t* a, const uint32_t (&b0)[L], uint32_t* m_row, uint32_t* dump,
                           const uint32_t (&nbar)[L], const Lanes<64>& ln, int rows) {
    constexpr int G = 64, kT = AbTrip<L>::kSteps;
    const uint32_t dmask = kLimbMask & ln.not_top;
    uint64_t p[L];
#pragma unroll
    for (int k = 0; k < L; ++k) p[k] = 0;
    wave::lds_u32* const w = wave::as_lds(ln.g == 0u ? m_row : dump + 4 * ln.g + 8);
    const int trips = (rows + kT) / kT;  // ceil((rows + 1) / kT) >= 2 (key_setup.h build_quick)
    PHE_BOUNDS(trips >= 2 && (trips + 1) * kT <= G * L + 48);  // digit blocks are fetched up to two trips ahead: still inside the slot
    uint32_t blk_a[kT], blk_b[kT], held[kT - 1];
#pragma unroll
    for (int i = 0; i + 1 < kT; ++i) held[i] = 0u;
    ab_load_block<kT>(blk_a, a);
    ab_load_block<kT>(blk_b, a + kT);
    ab_first_steps<L, kT, true, false>(p, blk_a, w, held, b0, nbar, dmask, ln);
    int t = 1;
#pragma unroll 1
    for (; t + 2 < trips; t += 2) {  // two full trips: the digit blocks change roles, nothing is copied
        ab_load_block<kT>(blk_a, a + (t + 1) * kT);
        ab_first_steps<L, kT, false, false>(p, blk_b, w + t * kT, held, b0, nbar, dmask, ln);
        ab_load_block<kT>(blk_b, a + (t + 2) * kT);
        ab_first_steps<L, kT, false, false>(p, blk_a, w + (t + 1) * kT, held, b0, nbar, dmask, ln);
    }
    if (t + 1 < trips) {
        ab_load_block<kT>(blk_a, a + (t + 1) * kT);
        ab_first_steps<L, kT, false, false>(p, blk_b, w + t * kT, held, b0, nbar, dmask, ln);
        ++t;
#pragma unroll
        for (int i = 0; i < kT; ++i) blk_b[i] = blk_a[i];
    }
    {   // the last trip: its last step takes no digit
        const int left = rows + 1 - t * kT;
        wave::lds_u32* const wl = w + t * kT;
        if constexpr (L == 1) {
            switch (left) {
                case 1: ab_first_steps<L, 1, false, true>(p, blk_b, wl, held, b0, nbar, dmask, ln); break;
                case 2: ab_first_steps<L, 2, false, true>(p, blk_b, wl, held, b0, nbar, dmask, ln); break;
                case 3: ab_first_steps<L, 3, false, true>(p, blk_b, wl, held, b0, nbar, dmask, ln); break;
                default: ab_first_steps<L, 4, false, true>(p, blk_b, wl, held, b0, nbar, dmask, ln); break;
            }
        } else {  // rows is a multiple of L: 1 or L + 1 steps
            if (left == 1) ab_first_steps<L, 1, false, true>(p, blk_b, wl, held, b0, nbar, dmask, ln);
            else ab_first_steps<L, L + 1, false, true>(p, blk_b, wl, held, b0, nbar, dmask, ln);
        }
    }
    normalize_partial<G, L>(z0, p, ln);
}

template <int L, int N, bool FIRST, bool LAST, bool MUL>
PHE_DEV void ab_second_steps(uint64_t (&q)[L], const uint32_t (&dig_a)[AbTrip<L>::kSteps], const uint32_t (&dig_c)[AbTrip<L>::kSteps],
                             const uint32_t (&m_cur)[AbTrip<L>::kSteps], const uint32_t (&b0)[L], const uint32_t (&b1)[L],
                             const uint32_t (&nbar)[L], uint32_t dmask, const Lanes<64>& ln) {
    constexpr int G = 64;
    constexpr bool kNarrow = (MUL ? 3 : 2) * L <= 7 && !PHE_AB_GENERIC_STEPS;  // see ab_shift_narrow
    // the lanes other than lane 0 read zeros where lane 0 reads the quotient row (ab_second_word): the digit of Q is added as it
    // comes — with narrow carries it rides on the carry's addition
#pragma unroll
    for (int jj = 0; jj < N; ++jj) {
        const bool quot = !(FIRST && jj == 0), digit = !(LAST && jj == N - 1);
        const int j = jj % L, jl = (jj + L - 1) % L;
        const uint32_t dm = digit ? m_cur[jj] : 0u;
        uint32_t m2 = 0;
        if (quot) {
            if constexpr (kNarrow) {
                const uint32_t t = wave::reread((uint32_t)q[jl] & kLimbMask);
                m2 = wave::grp_bcast0<G>(t, ln);
                ab_shift_narrow<L>(q, jl, t, dm);
            } else {
                m2 = bcast_digit<G>((uint32_t)q[jl], 0u, ln);
                shift_row<G, L>(q, jl, dmask);
                q[j] += (uint64_t)dm;
            }
        } else {
            q[j] += (uint64_t)dm;
        }
        if (digit) {
#pragma unroll
            for (int k = 0; k < L; ++k) q[(k + j) % L] = wave::mad64(dig_a[jj], b1[k], q[(k + j) % L]);
            if constexpr (MUL) {
#pragma unroll
                for (int k = 0; k < L; ++k) q[(k + j) % L] = wave::mad64(dig_c[jj], b0[k], q[(k + j) % L]);
            }
            if constexpr (L == 1) q[0] = wave::reread64(q[0]);  // (the digits' products first: the quotient digit is still on its way)
        }
        if (quot) {
#pragma unroll
            for (int k = 0; k < L; ++k) q[(k + j) % L] = wave::mad64(m2, nbar[k], q[(k + j) % L]);
        }
    }
}

// z1 = (Q + a*b1 [+ c*b0] + Q2*n~) / R with the digits of a, c and of the first word's quotient Q in LDS
// (zeros: H + 48 zero words of LDS: what the lanes above lane 0 read in place of the quotient row)
template <int L, bool MUL>
PHE_DEV void ab_second_word(uint32_t (&z1)[L], const uint32_t* a, const uint32_t* c, uint32_t* m_row, uint32_t* zeros,
                            const uint32_t (&b0)[L], const uint32_t (&b1)[L], const uint32_t (&nbar)[L], const Lanes<64>& ln, int rows) {
    constexpr int G = 64, kT = AbTrip<L>::kSteps;
    const uint32_t dmask = kLimbMask & ln.not_top;
    uint64_t q[L];
#pragma unroll
    for (int k = 0; k < L; ++k) q[k] = 0;
    const int trips = (rows + kT) / kT;
    const wave::lds_u32* const mz = wave::as_lds(ln.g == 0u ? m_row : zeros);
    uint32_t a_a[kT], a_b[kT], c_a[kT], c_b[kT], m_a[kT], m_b[kT];
    const auto load = [&](uint32_t (&da)[kT], uint32_t (&dc)[kT], uint32_t (&dm)[kT], int t) {
        ab_load_block<kT>(da, a + t * kT);
        ab_load_block<kT>(dm, mz + t * kT);
        if constexpr (MUL) ab_load_block<kT>(dc, c + t * kT);
    };
#pragma unroll
    for (int i = 0; i < kT; ++i) c_a[i] = c_b[i] = 0u;
    load(a_a, c_a, m_a, 0);
    load(a_b, c_b, m_b, 1);
    ab_second_steps<L, kT, true, false, MUL>(q, a_a, c_a, m_a, b0, b1, nbar, dmask, ln);
    int t = 1;
#pragma unroll 1
    for (; t + 2 < trips; t += 2) {  // two full trips: the digit blocks change roles, nothing is copied
        load(a_a, c_a, m_a, t + 1);
        ab_second_steps<L, kT, false, false, MUL>(q, a_b, c_b, m_b, b0, b1, nbar, dmask, ln);
        load(a_b, c_b, m_b, t + 2);
        ab_second_steps<L, kT, false, false, MUL>(q, a_a, c_a, m_a, b0, b1, nbar, dmask, ln);
    }
    if (t + 1 < trips) {
        load(a_a, c_a, m_a, t + 1);
        ab_second_steps<L, kT, false, false, MUL>(q, a_b, c_b, m_b, b0, b1, nbar, dmask, ln);
        ++t;
#pragma unroll
        for (int i = 0; i < kT; ++i) {
            a_b[i] = a_a[i];
            c_b[i] = c_a[i];
            m_b[i] = m_a[i];
        }
    }
    {   // the last trip: its last step takes no digits
        const int left = rows + 1 - t * kT;
        if constexpr (L == 1) {
            switch (left) {
                case 1: ab_second_steps<L, 1, false, true, MUL>(q, a_b, c_b, m_b, b0, b1, nbar, dmask, ln); break;
                case 2: ab_second_steps<L, 2, false, true, MUL>(q, a_b, c_b, m_b, b0, b1, nbar, dmask, ln); break;
                case 3: ab_second_steps<L, 3, false, true, MUL>(q, a_b, c_b, m_b, b0, b1, nbar, dmask, ln); break;
                default: ab_second_steps<L, 4, false, true, MUL>(q, a_b, c_b, m_b, b0, b1, nbar, dmask, ln); break;
            }
        } else {
            if (left == 1) ab_second_steps<L, 1, false, true, MUL>(q, a_b, c_b, m_b, b0, b1, nbar, dmask, ln);
            else ab_second_steps<L, L + 1, false, true, MUL>(q, a_b, c_b, m_b, b0, b1, nbar, dmask, ln);
        }
    }
    normalize_partial<G, L>(z1, q, ln);
}

// The batched exponentiation of modexp_split_body for ONE number on a wave pair: role 0 = first words (A), 1 = second (B).
// lds: ab_lds_words<L>() words of the workgroup (two slots of a-digits | m-digits, B's own digit row, one row for words handed
// over, the area for the other lanes' quotient-digit copies);
// tbl: (tbl_entries + 1) * 2H words for this number's window table (the extra entry carries base^2 during the table build);
// sched: the exponent's schedule (A.sched, or a copy of it).  Both live in LDS where they fit (split_kernels.inc): a global
// load per window would be waited for at the very next workgroup barrier (its release fence drains the memory counters),
// i.e. a full memory latency per window — as long as three products of a 2048-bit key.
constexpr int kAbTableEntries = 33;   // LDS window table: 2^5 odd powers + base^2 (windows of up to 6 bits)
constexpr int kAbSchedWords = 1024;   // LDS copy of the schedule
template <int L, int MODE>
PHE_DEV void modexp_split_ab_body(const SplitArgs& A, uint32_t* lds, uint32_t* tbl, const uint32_t* sched, uint64_t item, bool live,
                                  uint32_t role, uint32_t lane) {
    constexpr int G = 64, H = G * L, S2 = 2 * H;
    const Lanes<G> ln(lane);
    const uint32_t g = ln.g;
    const bool first = role == 0u;
    uint32_t nbar[L];  // (n~ + 1) / 2^29 of the scaled modulus: what the sweeps multiply the quotient digits by
    load_row<L>(nbar, A.mod.nbar, g);
    const int rows = A.mod.rows;
    uint32_t* own_c = lds + 4 * H;  // B: digits of its word X1 (the c operand of a product)
    uint32_t* mail = lds + 5 * H;   // a word handed from one role to the other
    uint32_t* dump = lds + 6 * H;   // ab_first_word: where the lanes other than lane 0 leave their copies of the quotient digits
    uint32_t* zeros = dump + 4 * G + H + 16;  // ab_second_word: H + 48 zero words
    if (!first) {
#pragma unroll
        for (int t = 0; t < L; ++t) zeros[g * L + t] = 0u;
        if (g < 48u) zeros[H + g] = 0u;
        wave::lds_fence();
    }
    int k = 0;                      // products so far: slot k & 1
    const auto slot_a = [&](int kk) { return lds + (kk & 1) * S2; };
    const auto slot_m = [&](int kk) { return lds + (kk & 1) * S2 + H; };
    uint32_t W[L], V[L], Y0[L];     // this role's word of X; of the factor Y; (B only) the FIRST word of Y
    // one product step: A publishes `dig` (the multiplier's first word) and sweeps it against b0; B, one barrier later, sweeps
    // the second word.  MUL: the factor is the pair (Y0, V); else the multiplier's own pair (squaring) or a plain chunk.
    const auto step_plain = [&](const uint32_t (&dig)[L], const uint32_t (&b)[L], uint32_t (&out)[L]) {
        if (first) {
            lds_put<L>(slot_a(k), dig, g);
            ab_first_word<L>(out, slot_a(k), b, slot_m(k), dump, nbar, ln, rows);
            wave::block_barrier();
        } else {
            wave::block_barrier();
            ab_second_word<L, false>(out, slot_a(k), nullptr, slot_m(k), zeros, b, b, nbar, ln, rows);
        }
        ++k;
    };
    const auto square = [&]() {
        if (first) {
            step_plain(W, W, W);
        } else {
            uint32_t d[L];
#pragma unroll
            for (int t = 0; t < L; ++t) d[t] = W[t];
            add_normalize<G, L>(d, W, ln);  // 2*X1
            step_plain(d, d, W);
        }
    };
    // X <- X*Y: A has (W = X0, V = Y0), B has (W = X1, V = Y1) and needs Y0 too.  y0_late: where B reads Y0 AFTER the barrier
    // — for a Y that A has only just stored (its store precedes A's arrival at this barrier); nullptr: B holds Y0 already
    const auto multiply = [&](const uint32_t* y0_late) {
        if (first) {
            lds_put<L>(slot_a(k), W, g);
            ab_first_word<L>(W, slot_a(k), V, slot_m(k), dump, nbar, ln, rows);
            wave::block_barrier();
        } else {
            lds_put<L>(own_c, W, g);
            wave::block_barrier();
            if (y0_late) load_row<L>(Y0, y0_late, g);
            ab_second_word<L, true>(W, slot_a(k), own_c, slot_m(k), zeros, Y0, V, nbar, ln, rows);
        }
        ++k;
    };
    const auto load_entry = [&](uint32_t (&dst)[L], int e, uint32_t word) { load_row<L>(dst, tbl + (size_t)e * S2 + word * H, g); };
    const auto store_entry = [&](int e) { store_row<L>(tbl + (size_t)e * S2 + role * H, W, g); };

    // ---- the base into the pair form (split_conv): chunk j times pair(R^(j+2)), summed; one product with pair(1) if summed ----
    {
        const uint32_t* src = A.base + item * (uint64_t)A.base_limbs;
        for (int j = 0; j < A.base_chunks; ++j) {
            uint32_t dig[L], cst[L], t[L];
            load_u32_as_r29<L>(dig, src, A.base_limbs, j * rows, g, rows);
            load_row<L>(cst, A.mod.conv + (size_t)(2 * j + (int)role) * H, g);
            if (j == 0) {
                step_plain(dig, cst, W);
            } else {
                step_plain(dig, cst, t);
                add_normalize<G, L>(W, t, ln);
            }
        }
        if (A.base_chunks > 1) {
            load_row<L>(V, A.mod.e + role * H, g);
            if (!first) load_row<L>(Y0, A.mod.e, g);
            multiply(nullptr);
        }
    }
    // ---- odd powers base^1, base^3, ... ----------------------------------------------------------------------------
    store_entry(0);
    if (A.tbl_entries > 1) {
        const int extra = A.tbl_entries;  // base^2: its first word travels to B through this entry
        uint32_t keep[L];
#pragma unroll
        for (int t = 0; t < L; ++t) keep[t] = W[t];
        square();
        store_entry(extra);
#pragma unroll
        for (int t = 0; t < L; ++t) {
            V[t] = W[t];
            W[t] = keep[t];
        }
        for (int j = 1; j < A.tbl_entries; ++j) {
            multiply(j == 1 ? tbl + (size_t)extra * S2 : nullptr);  // base^2's first word: A stored it before it came to this barrier
            store_entry(j);
        }
    }
    // ---- left-to-right sliding window ------------------------------------------------------------------------------
    // B reads A's word of a table entry (Y0) at the top of a window: every entry must be in place before the first one is
    // asked for.  All but the last stored were followed by a product's barrier; this one covers the last (with a table of one
    // entry: the only one).  On the device B is a product's worth of time behind A's store anyway; the emulator's two host
    // threads are not (tests/test_emu_core.py::test_scalar_multiplication_of_a_handful_on_wave_pairs found it with exponents of
    // a few bits).
    wave::block_barrier();
    load_entry(W, A.first_idx, role);
    for (int op = 0; op < A.n_ops; ++op) {
        const uint32_t w = wave::grp_bcast0<G>(sched[op], ln);  // (wave-uniform: the loops below are scalar loops)
        const int nsq = (int)(w >> 8), sel = (int)(w & 0xffu);
        if (sel) {  // the factor's words are fetched before the squarings: they arrive under them
            load_entry(V, sel - 1, role);
            if (!first) load_entry(Y0, sel - 1, 0u);
        }
        for (int sq = 0; sq < nsq; ++sq) square();
        if (sel) multiply(nullptr);
    }
    // ---- the way out: both words with A ------------------------------------------------------------------------------
    if (!first) lds_put<L>(mail, W, g);
    wave::block_barrier();
    if (first) {
        uint32_t X1[L];
        load_row<L>(X1, mail, g);
        SplitLane<G, L, false> K;  // the way out works modulo the true n (same R): X0 - n~*X1 = X0 - n*(k*X1)
        load_row<L>(K.n, A.exit_mod.n, g);
        K.n0inv = A.exit_mod.n0inv;
        K.rows_ = rows;
        K.row_a = lds;
        K.row_c = lds + H;
        const uint32_t* mp = nullptr;
        if (MODE == kModeEncrypt) mp = A.post ? A.post + item * (uint64_t)A.post_limbs : nullptr;
        split_exit<G, L>(A.out + item * (uint64_t)A.out_limbs, A.out_limbs, W, X1, mp, A.post_limbs, A.exit_mod, K, ln, live,
                         A.exit_mod.kx);
    }
}

// Element-wise product modulo n^2 on the pair representation, for key widths whose n^2 has no full-width geometry
// (mont_core.h stops at 16 x 18 = 288 limbs ~ 8344 bits, i.e. keys up to ~4170 bits; examples/benchmarks.py:88-90 of the
// reference times 8192-bit keys): both factors enter the pair form, one pair product, the canonical residue leaves
// (same MulArgs as mulmod_body; b_plain_limbs > 0: a * (1 + n*m), the plaintext folded into split_exit as in encrypt).
struct SplitMulArgs {
    SplitConsts mod;
    const uint32_t* a;
    const uint32_t* b;
    uint32_t* out;
    size_t a_stride, b_stride, out_stride;
    int limbs;          // 32-bit words of a ciphertext row
    int chunks;         // ceil(32*limbs / (29 H))
    int b_plain_limbs;  // 0: b is a residue mod n^2; > 0: b is a plaintext of that many words
    uint64_t batch;
};

template <int G, int L>
PHE_DEV void mulmod_split_body(const SplitMulArgs& A, uint32_t* lds_row, uint32_t slot, uint32_t total_slots, uint32_t lane) {
    constexpr int H = G * L;
    const Lanes<G> ln(lane);
    const uint32_t g = ln.g;
    SplitLane<G, L> K;
    load_row<L>(K.n, A.mod.n, g);
    K.n0inv = A.mod.n0inv;
    K.rows_ = A.mod.rows;
    K.row_a = lds_row;
    K.row_c = lds_row + H;
    const uint64_t n_iter = (A.batch + total_slots - 1) / total_slots;
    for (uint64_t it = 0; it < n_iter; ++it) {
        uint64_t item = slot + it * (uint64_t)total_slots;
        const bool live = item < A.batch;
        if (!live) item = A.batch - 1;
        uint32_t X0[L], X1[L];
        split_conv<G, L>(X0, X1, A.a + item * A.a_stride, A.limbs, A.chunks, A.mod, K, ln);
        const uint32_t* mp = nullptr;
        if (A.b_plain_limbs > 0) {
            mp = A.b + item * A.b_stride;
        } else {
            uint32_t Y0[L], Y1[L];
            split_conv<G, L>(Y0, Y1, A.b + item * A.b_stride, A.limbs, A.chunks, A.mod, K, ln);
            split_mul<G, L>(X0, X1, Y0, Y1, K, ln);
        }
        split_exit<G, L>(A.out + item * A.out_stride, A.limbs, X0, X1, mp, A.b_plain_limbs, A.mod, K, ln, live);
    }
}

// ---- resident ciphertext rows in the pair form ("the engine's own format") ---------------------------------------------
// A device-resident ciphertext vector that is only multiplied (chains of EncryptedNumber.__add__ = _raw_add,
// phe/paillier.py:705-719; sum() trees; the obfuscators r^n made ahead of time) need not go through 32-bit words between
// two steps: a row is kept as (X0 | X1), 2H almost-normalised 29-bit limbs in global limb order — 144 words for a 2048-bit
// key's 128-word ciphertext.  A homomorphic addition on such rows is ONE pair product (5*H^2 multiply-adds; the full-width
// form costs two Montgomery products = 16*H^2, its one-product "debt" form 8*H^2) with no slicing of words into limbs, no
// carry look-ahead and no conditional subtraction, and the pair form of a product IS the product's pair form: nothing to
// settle.  to_pair_body / from_pair_body convert at the boundary (split_conv / split_exit, the very routines every
// exponentiation enters and leaves by), so what reaches the caller is the same canonical residue as ever.
struct PairArgs {
    SplitConsts mod;
    const uint32_t* a;    // to_pair: (batch, limbs) 32-bit-word rows;  from_pair / pair_mul: (batch, 2H) pair rows
    const uint32_t* b;    // pair_mul: pair rows, b_stride words apart (0: one row for the whole batch);
                          // from_pair: plaintexts (batch, b_limbs) to fold in as 1 + n*m, or nullptr
    uint32_t* out;        // to_pair / pair_mul: (batch, 2H) pair rows;  from_pair: (batch, limbs) 32-bit-word rows
    size_t b_stride;
    int limbs;            // 32-bit words of a ciphertext row
    int chunks;           // ceil(32*limbs / (29 H))
    int b_limbs;
    uint64_t batch;
};

template <int G, int L>
PHE_DEV void to_pair_body(const PairArgs& A, uint32_t* lds_row, uint32_t slot, uint32_t total_slots, uint32_t lane) {
    constexpr int H = G * L, S2 = 2 * H;
    const Lanes<G> ln(lane);
    const uint32_t g = ln.g;
    SplitLane<G, L> K;
    load_row<L>(K.n, A.mod.n, g);
    K.n0inv = A.mod.n0inv;
    K.rows_ = A.mod.rows;
    K.row_a = lds_row;
    K.row_c = lds_row + H;
    const uint64_t n_iter = (A.batch + total_slots - 1) / total_slots;
    for (uint64_t it = 0; it < n_iter; ++it) {
        uint64_t item = slot + it * (uint64_t)total_slots;
        const bool live = item < A.batch;
        if (!live) item = A.batch - 1;
        uint32_t X0[L], X1[L];
        split_conv<G, L>(X0, X1, A.a + item * (uint64_t)A.limbs, A.limbs, A.chunks, A.mod, K, ln);
        if (live) {
            store_row<L>(A.out + item * (uint64_t)S2, X0, g);
            store_row<L>(A.out + item * (uint64_t)S2 + H, X1, g);
        }
    }
}

template <int G, int L>
PHE_DEV void from_pair_body(const PairArgs& A, uint32_t* lds_row, uint32_t slot, uint32_t total_slots, uint32_t lane) {
    constexpr int H = G * L, S2 = 2 * H;
    const Lanes<G> ln(lane);
    const uint32_t g = ln.g;
    SplitLane<G, L> K;
    load_row<L>(K.n, A.mod.n, g);
    K.n0inv = A.mod.n0inv;
    K.rows_ = A.mod.rows;
    K.row_a = lds_row;
    K.row_c = lds_row + H;
    const uint64_t n_iter = (A.batch + total_slots - 1) / total_slots;
    for (uint64_t it = 0; it < n_iter; ++it) {
        uint64_t item = slot + it * (uint64_t)total_slots;
        const bool live = item < A.batch;
        if (!live) item = A.batch - 1;
        uint32_t X0[L], X1[L];
        load_row<L>(X0, A.a + item * (uint64_t)S2, g);
        load_row<L>(X1, A.a + item * (uint64_t)S2 + H, g);
        const uint32_t* mp = A.b ? A.b + item * (uint64_t)A.b_limbs : nullptr;
        split_exit<G, L>(A.out + item * (uint64_t)A.limbs, A.limbs, X0, X1, mp, A.b_limbs, A.mod, K, ln, live);
    }
}

template <int G, int L>
PHE_DEV void pair_mul_body(const PairArgs& A, uint32_t* lds_row, uint32_t slot, uint32_t total_slots, uint32_t lane) {
    constexpr int H = G * L, S2 = 2 * H;
    const Lanes<G> ln(lane);
    const uint32_t g = ln.g;
    SplitLane<G, L> K;
    load_row<L>(K.n, A.mod.n, g);
    K.n0inv = A.mod.n0inv;
    K.rows_ = A.mod.rows;
    K.row_a = lds_row;
    K.row_c = lds_row + H;
    const uint64_t n_iter = (A.batch + total_slots - 1) / total_slots;
    for (uint64_t it = 0; it < n_iter; ++it) {
        uint64_t item = slot + it * (uint64_t)total_slots;
        const bool live = item < A.batch;
        if (!live) item = A.batch - 1;
        uint32_t X0[L], X1[L], Y0[L], Y1[L];
        load_row<L>(X0, A.a + item * (uint64_t)S2, g);
        load_row<L>(X1, A.a + item * (uint64_t)S2 + H, g);
        load_row<L>(Y0, A.b + item * A.b_stride, g);
        load_row<L>(Y1, A.b + item * A.b_stride + H, g);
        split_mul<G, L>(X0, X1, Y0, Y1, K, ln);
        if (live) {
            store_row<L>(A.out + item * (uint64_t)S2, X0, g);
            store_row<L>(A.out + item * (uint64_t)S2 + H, X1, g);
        }
    }
}

// Encryption by the key owner: r^n mod n^2 from its two CRT halves (phe/paillier.py:137 obfuscator = powmod(r, n, nsquare)
// is one exponentiation modulo n^2; whoever holds p and q can take it modulo p^2 and modulo q^2 — half-width numbers, a
// quarter of the multiply-adds each — and lift).  With y_p = r^n mod p^2, y_q = r^n mod q^2 (canonical, from the
// half-exponentiation kernels run with the exponent n) and K = (p^2)^-1 mod q^2:
//     u = (y_q - y_p) * K mod q^2 = y_q*K + y_p*(q^2 - K)   (mod q^2: additions only),      y = y_p + p^2 * u  <  n^2
// which IS the canonical residue r^n mod n^2: the same bits the one big exponentiation returns.  Full-width geometry of
// q^2: S = G*L limbs, R = 2^(29 S) >= 16 q^2.
struct CrtLiftArgs {
    ModConsts mod;        // modulus q^2 (n, r1 used)
    const uint32_t* kr;   // K * R mod q^2
    const uint32_t* nkr;  // (q^2 - K) * R mod q^2
    const uint32_t* psq;  // p^2
    const uint32_t* yp;   // (batch, x_stride) words
    const uint32_t* yq;
    size_t x_stride;
    int x_limbs;          // words of a row that hold the value
    uint32_t* out;        // (batch, out_limbs) words: r^n mod n^2
    int out_limbs;
    uint64_t batch;
};

template <int G, int L>
PHE_DEV void crt_lift_body(const CrtLiftArgs& A, uint32_t* lds_row, uint32_t slot, uint32_t total_slots, uint32_t lane) {
    constexpr int S = G * L;
    const Lanes<G> ln(lane);
    const uint32_t g = ln.g;
    const uint32_t n0inv = A.mod.n0inv;
    uint32_t* row = lds_row;       // S digits: the multiplier of the current product
    uint32_t* row2 = lds_row + S;  // S digits: the low half of the final product
    uint32_t n[L];
    load_row<L>(n, A.mod.n, g);
    const uint64_t n_iter = (A.batch + total_slots - 1) / total_slots;
    for (uint64_t it = 0; it < n_iter; ++it) {
        uint64_t item = slot + it * (uint64_t)total_slots;
        const bool live = item < A.batch;
        if (!live) item = A.batch - 1;
        uint32_t a[L], m1[L], m2[L], cst[L];
        load_u32_as_r29<L>(a, A.yq + item * A.x_stride, A.x_limbs, 0, g);
        lds_put<L>(row, a, g);
        load_row<L>(cst, wave::reread_ptr(A.kr), g);
        montmul<G, L>(m1, row, cst, n, n0inv, ln);  // y_q * K
        load_u32_as_r29<L>(a, A.yp + item * A.x_stride, A.x_limbs, 0, g);
        lds_put<L>(row, a, g);
        load_row<L>(cst, wave::reread_ptr(A.nkr), g);
        montmul<G, L>(m2, row, cst, n, n0inv, ln);  // y_p * (q^2 - K)
        add_normalize<G, L>(m1, m2, ln);            // < 4 q^2
        lds_put<L>(row, m1, g);
        load_row<L>(cst, wave::reread_ptr(A.mod.r1), g);
        montmul<G, L>(m1, row, cst, n, n0inv, ln);  // the same value, below 2 q^2 again
        canonicalize<G, L>(m1, n, ln);              // u in [0, q^2)
        lds_put<L>(row, m1, g);
        load_row<L>(cst, wave::reread_ptr(A.psq), g);
        uint32_t hi[L], lo[L];
        mul_wide<G, L>(hi, row, cst, a, row2, ln);  // u * p^2 + y_p = hi * R + lo
        wave::lds_fence();
        load_row<L>(lo, row2, g);
        normalize_full<G, L>(hi, ln);
        store_pair_as_u32<G, L>(A.out + item * (uint64_t)A.out_limbs, A.out_limbs, lo, hi, row, g, live);
    }
}

// Per-element exponents (phe/paillier.py:751 powmod(c, scalar, n^2); :749 with the inverted base) on the pair
// representation: fixed 2^w-ary windows over the batch-wide maximum bit length, as modexp_var_body (mont_core.h).
struct SplitVarArgs {
    SplitConsts mod;
    const uint32_t* base;  // (batch, base_limbs), any value below 2^(32*base_limbs)
    int base_limbs;
    int base_chunks;
    const uint32_t* exps;  // (batch, exp_limbs)
    int exp_limbs;
    int window;      // w in 1..6 (key_setup.h:pick_window)
    int n_windows;   // ceil(max_bits / w), >= 1
    uint32_t* out;
    int out_limbs;
    uint32_t* table;  // scratch: total_groups * 2^w * 2H words
    uint64_t batch;
    int pair_io;      // 1: base and out are rows in the pair form (2H limbs each, "resident rows in the pair form"): no conversion
                      // in (13 H^2 multiply-adds) and no exit (8 H^2) — the PAIR instantiation of the kernel
};

template <int G, int L, bool PAIR = false>
PHE_DEV void modexp_var_split_body(const SplitVarArgs& A, uint32_t* lds_row, uint32_t slot, uint32_t total_slots,
                                   uint32_t lane) {
    constexpr int H = G * L, S2 = 2 * H;
    const Lanes<G> ln(lane);
    const uint32_t g = ln.g;
    SplitLane<G, L> K;
    load_row<L>(K.n, A.mod.n, g);
    K.n0inv = A.mod.n0inv;
    K.rows_ = A.mod.rows;
    K.row_a = lds_row;
    K.row_c = lds_row + H;
    const int tbl_entries = 1 << A.window;
    uint32_t* tbl = A.table + (size_t)slot * (size_t)tbl_entries * S2;
    const uint64_t n_iter = (A.batch + total_slots - 1) / total_slots;
    for (uint64_t it = 0; it < n_iter; ++it) {
        uint64_t item = slot + it * (uint64_t)total_slots;
        const bool live = item < A.batch;
        if (!live) item = A.batch - 1;
        uint32_t X0[L], X1[L], Y0[L], Y1[L];
        if constexpr (PAIR) {
            load_row<L>(Y0, A.base + item * (uint64_t)S2, g);
            load_row<L>(Y1, A.base + item * (uint64_t)S2 + H, g);
        } else {
            split_conv<G, L>(Y0, Y1, A.base + item * (uint64_t)A.base_limbs, A.base_limbs, A.base_chunks, A.mod, K, ln);
        }
        // table: base^0 (the pair of 1) .. base^(2^w - 1)
        load_row<L>(X0, wave::reread_ptr(A.mod.e), g);
        load_row<L>(X1, wave::reread_ptr(A.mod.e) + H, g);
        store_row<L>(tbl, X0, g);
        store_row<L>(tbl + H, X1, g);
        store_row<L>(tbl + S2, Y0, g);
        store_row<L>(tbl + S2 + H, Y1, g);
#pragma unroll
        for (int k = 0; k < L; ++k) {
            X0[k] = Y0[k];
            X1[k] = Y1[k];
        }
        for (int j = 2; j < tbl_entries; ++j) {
            split_mul<G, L>(X0, X1, Y0, Y1, K, ln);
            store_row<L>(tbl + (size_t)j * S2, X0, g);
            store_row<L>(tbl + (size_t)j * S2 + H, X1, g);
        }
        const uint32_t* e = A.exps + item * (uint64_t)A.exp_limbs;
        uint32_t d = exp_digit(e, A.exp_limbs, (A.n_windows - 1) * A.window, A.window);
        load_row<L>(X0, tbl + (size_t)d * S2, g);
        load_row<L>(X1, tbl + (size_t)d * S2 + H, g);
        for (int wi = A.n_windows - 2; wi >= 0; --wi) {
            for (int s = 0; s < A.window; ++s) split_square<G, L>(X0, X1, K, ln);
            // (the exponent row, the table and the lane position are re-read where they are used: left to itself the compiler
            //  keeps their derived addresses alive across the sweeps and spills other values inside this loop for them)
            d = exp_digit(wave::reread_vptr(e), A.exp_limbs, wi * A.window, A.window);
            if (wave::ballot(d != 0) != 0) {  // wave-uniform: skip when every group has a zero digit
                const uint32_t* entry = wave::reread_vptr(tbl) + (size_t)d * S2;
                const uint32_t gi = wave::reread(g);
                load_row<L>(Y0, entry, gi);
                load_row<L>(Y1, entry + H, gi);
                split_mul<G, L>(X0, X1, Y0, Y1, K, ln);
            }
        }
        if constexpr (PAIR) {
            if (live) {
                store_row<L>(A.out + item * (uint64_t)S2, X0, g);
                store_row<L>(A.out + item * (uint64_t)S2 + H, X1, g);
            }
        } else {
            split_exit<G, L>(A.out + item * (uint64_t)A.out_limbs, A.out_limbs, X0, X1, nullptr, 0, A.mod, K, ln, live);
        }
    }
}

// Multi-exponentiation  prod_i base_i ^ e_i  mod n^2 — the encrypted dot product sum_i k_i * E(x_i)
// (phe/tests/math_test.py:44-58 np.dot over EncryptedNumbers; examples/logistic_regression_encrypted_model.py:170-177;
// each term is phe/paillier.py:751 powmod(c_i, k_i, n^2), the terms are joined by :705-719 mulmod).  The product is a
// canonical residue, so it does not depend on the order of the factors: one limb group takes a CHUNK of elements and
// runs them through ONE square-and-multiply ladder (Straus' interleaving) — the w squarings per window are shared
// by the whole chunk, each element contributes one product per window from its own 2^w-ary table.  Per element
// that is (bits/chunk) squarings + bits/w + 2^w - 2 products instead of bits squarings + bits/w + 2^w - 2 products.
//
// Matrix form (a plaintext matrix times an encrypted vector: `rows` dot products over the same ciphertexts): the
// exponents are a (rows, batch) matrix, a task is (chunk j, block of row_block rows) and runs one ladder per row of
// its block on the tables it built once.  Scalars on the reference's negative branch (phe/paillier.py:745-749:
// inverted base, exponent n - k) differ from row to row, so the inverted ciphertexts come as a second base array
// with their own tables and a (rows, batch) byte mask selects per entry.
// out[(j * rows + r)] = product over the elements of chunk j with the exponents of row r; the caller joins the
// chunks of every row with a pairwise k_mulmod tree (contiguous: the chunk index is the slow one).
struct SplitMultiArgs {
    SplitConsts mod;
    const uint32_t* base;      // (batch, base_limbs)
    const uint32_t* base_inv;  // (batch, base_limbs) inverses mod n^2, or nullptr (then no entry is negative)
    int base_limbs;
    int base_chunks;
    const uint32_t* exps;  // (rows, batch, exp_limbs)
    const uint8_t* neg;    // (rows, batch): nonzero = take the inverted base; nullptr = none
    int exp_limbs;
    int window;     // w in 1..4
    int n_windows;  // ceil(max_bits / w), >= 1
    int chunk;      // elements per limb group and ladder
    int row_block;  // rows a task runs on its tables
    uint32_t* out;  // (n_chunks, rows, out_limbs)
    int out_limbs;
    uint32_t* table;  // scratch: total_groups * chunk * (2^w - 1) * (base_inv ? 2 : 1) * 2H words
    uint64_t batch;
    uint64_t rows;
    uint64_t n_chunks;      // ceil(batch / chunk)
    uint64_t n_row_blocks;  // ceil(rows / row_block)
    int pair_in;            // 1: base (and base_inv) are resident rows in the pair form (2H limbs each, base_limbs = 2H): no conversion
                            // in — 13 H^2 of the ~170 H^2 multiply-adds an element of a 56-bit dot product costs
};

template <int G, int L>
PHE_DEV void multiexp_split_body(const SplitMultiArgs& A, uint32_t* lds_row, uint32_t slot, uint32_t total_slots,
                                 uint32_t lane) {
    constexpr int H = G * L, S2 = 2 * H;
    const Lanes<G> ln(lane);
    const uint32_t g = ln.g;
    SplitLane<G, L> K;
    load_row<L>(K.n, A.mod.n, g);
    K.n0inv = A.mod.n0inv;
    K.rows_ = A.mod.rows;
    K.row_a = lds_row;
    K.row_c = lds_row + H;
    const int per = (1 << A.window) - 1;  // table of one element: base^1 .. base^(2^w - 1)
    const int signs = A.base_inv ? 2 : 1;
    uint32_t* tbl = A.table + (size_t)slot * (size_t)A.chunk * (size_t)per * (size_t)signs * S2;
    const uint64_t n_tasks = A.n_chunks * A.n_row_blocks;
    const uint64_t n_iter = (n_tasks + total_slots - 1) / total_slots;
    for (uint64_t it = 0; it < n_iter; ++it) {
        uint64_t task = slot + it * (uint64_t)total_slots;
        const bool live_task = task < n_tasks;
        if (!live_task) task = n_tasks - 1;
        const uint64_t cj = task % A.n_chunks, rb = task / A.n_chunks;
        const uint64_t first = cj * (uint64_t)A.chunk;
        // elements of this chunk that exist; every loop below runs `chunk` times in all groups so that the
        // wavefront stays converged — missing elements repeat the last one with digit 0
        const int count = (int)((A.batch - first < (uint64_t)A.chunk) ? A.batch - first : (uint64_t)A.chunk);
        uint32_t X0[L], X1[L], Y0[L], Y1[L];
        for (int sg = 0; sg < signs; ++sg) {
            const uint32_t* src = sg ? A.base_inv : A.base;
            for (int el = 0; el < A.chunk; ++el) {
                uint64_t item = first + (uint64_t)el;
                if (item >= A.batch) item = A.batch - 1;
                if (A.pair_in) {
                    load_row<L>(Y0, src + item * (uint64_t)A.base_limbs, g);
                    load_row<L>(Y1, src + item * (uint64_t)A.base_limbs + H, g);
                } else {
                    split_conv<G, L>(Y0, Y1, src + item * (uint64_t)A.base_limbs, A.base_limbs, A.base_chunks, A.mod, K, ln);
                }
                uint32_t* t = tbl + ((size_t)sg * (size_t)A.chunk + (size_t)el) * (size_t)per * S2;
                store_row<L>(t, Y0, g);
                store_row<L>(t + H, Y1, g);
#pragma unroll
                for (int k = 0; k < L; ++k) {
                    X0[k] = Y0[k];
                    X1[k] = Y1[k];
                }
                for (int j = 1; j < per; ++j) {
                    split_mul<G, L>(X0, X1, Y0, Y1, K, ln);
                    store_row<L>(t + (size_t)j * S2, X0, g);
                    store_row<L>(t + (size_t)j * S2 + H, X1, g);
                }
            }
        }
        for (int rr = 0; rr < A.row_block; ++rr) {
            uint64_t r = rb * (uint64_t)A.row_block + (uint64_t)rr;
            const bool live = live_task && r < A.rows;
            if (r >= A.rows) r = A.rows - 1;
            const uint64_t entry0 = r * A.batch + first;  // (row, first element of the chunk)
            load_row<L>(X0, A.mod.e, g);
            load_row<L>(X1, A.mod.e + H, g);
            for (int wi = A.n_windows - 1; wi >= 0; --wi) {
                if (wi != A.n_windows - 1)
                    for (int s = 0; s < A.window; ++s) split_square<G, L>(X0, X1, K, ln);
                for (int el = 0; el < A.chunk; ++el) {
                    uint32_t d = 0, sg = 0;
                    if (el < count) {
                        d = exp_digit(A.exps + (entry0 + (uint64_t)el) * (uint64_t)A.exp_limbs, A.exp_limbs, wi * A.window, A.window);
                        if (A.neg) sg = A.neg[entry0 + (uint64_t)el] ? 1u : 0u;
                    }
                    if (wave::ballot(d != 0) != 0) {  // wave-uniform: skip when every group has a zero digit
                        const uint32_t* src =
                            d ? tbl + (((size_t)sg * (size_t)A.chunk + (size_t)el) * (size_t)per + (d - 1)) * S2 : A.mod.e;
                        load_row<L>(Y0, src, g);
                        load_row<L>(Y1, src + H, g);
                        split_mul<G, L>(X0, X1, Y0, Y1, K, ln);
                    }
                }
            }
            split_exit<G, L>(A.out + (cj * A.rows + r) * (uint64_t)A.out_limbs, A.out_limbs, X0, X1, nullptr, 0, A.mod, K, ln, live);
        }
    }
}

// ---- multi-exponentiation on ONE table set for the whole vector ---------------------------------------------------
// A plaintext matrix with many rows, or sparse rows (examples/logistic_regression_encrypted_model.py:170-177 walks the
// NONZERO features of a sample: `_, idx = x.nonzero(); for i in idx: score += x[0, i] * self.weights[i]`), makes the
// per-task tables of multiexp_split_body a waste: the same few ciphertexts are expanded again and again.  Here the
// 2^w-ary tables of EVERY ciphertext of the vector (and of its inverse, if any entry is negative) are built once
// (multiexp_tables_body: one limb group per ciphertext) and a limb group then runs ONE ladder for a whole matrix row,
// over that row's entries only — (column, exponent, sign) triples in CSR order, or the dense row — with one table
// lookup and one pair product per entry and window.  The result of a row is final: no product tree.
// Rows are visited in the caller's `order` (rows sorted by their entry count) so that the 64/G groups of a wavefront
// have ladders of similar length; every inner loop runs to the wavefront's longest row and is predicated per group.
struct SplitTableArgs {
    SplitConsts mod;
    const uint32_t* base;      // (batch, base_limbs)
    const uint32_t* base_inv;  // (batch, base_limbs) or nullptr
    int base_limbs;
    int base_chunks;
    int window;       // w: the table of one ciphertext is base^1 .. base^(2^w - 1)
    uint32_t* table;  // (batch, signs, 2^w - 1, 2H)
    uint64_t batch;
};

template <int G, int L>
PHE_DEV void multiexp_tables_body(const SplitTableArgs& A, uint32_t* lds_row, uint32_t slot, uint32_t total_slots,
                                  uint32_t lane) {
    constexpr int H = G * L, S2 = 2 * H;
    const Lanes<G> ln(lane);
    const uint32_t g = ln.g;
    SplitLane<G, L> K;
    load_row<L>(K.n, A.mod.n, g);
    K.n0inv = A.mod.n0inv;
    K.rows_ = A.mod.rows;
    K.row_a = lds_row;
    K.row_c = lds_row + H;
    const int per = (1 << A.window) - 1;
    const int signs = A.base_inv ? 2 : 1;
    const uint64_t n_items = A.batch * (uint64_t)signs;
    const uint64_t n_iter = (n_items + total_slots - 1) / total_slots;
    for (uint64_t it = 0; it < n_iter; ++it) {
        uint64_t item = slot + it * (uint64_t)total_slots;
        const bool live = item < n_items;
        if (!live) item = n_items - 1;
        const uint64_t col = item / (uint64_t)signs;
        const int sg = (int)(item - col * (uint64_t)signs);
        const uint32_t* src = (sg ? A.base_inv : A.base) + col * (uint64_t)A.base_limbs;
        uint32_t X0[L], X1[L], Y0[L], Y1[L];
        split_conv<G, L>(Y0, Y1, src, A.base_limbs, A.base_chunks, A.mod, K, ln);
        uint32_t* t = A.table + item * (uint64_t)per * S2;
#pragma unroll
        for (int k = 0; k < L; ++k) {
            X0[k] = Y0[k];
            X1[k] = Y1[k];
        }
        for (int j = 0; j < per; ++j) {
            if (j) split_mul<G, L>(X0, X1, Y0, Y1, K, ln);
            if (live) {
                store_row<L>(t + (size_t)j * S2, X0, g);
                store_row<L>(t + (size_t)j * S2 + H, X1, g);
            }
        }
    }
}

struct SplitLookupArgs {
    SplitConsts mod;
    const uint32_t* table;    // multiexp_tables_body's output
    int signs;                // 1 or 2 (inverse tables present)
    const uint64_t* row_ptr;  // (rows + 1) entry offsets, or nullptr: dense rows of `batch` entries each
    const uint32_t* cols;     // (entries) column of every entry, or nullptr: dense (column = position in the row)
    const uint32_t* exps;     // (entries, exp_limbs)
    const uint8_t* neg;       // (entries) nonzero = inverted base, or nullptr
    const uint32_t* order;    // (rows) visiting order, or nullptr
    int exp_limbs;
    int window;
    int n_windows;
    uint32_t* out;  // (rows, out_limbs)
    int out_limbs;
    uint64_t batch;
    uint64_t rows;
};

template <int G, int L>
PHE_DEV void multiexp_lookup_body(const SplitLookupArgs& A, uint32_t* lds_row, uint32_t slot, uint32_t total_slots,
                                  uint32_t lane) {
    constexpr int H = G * L, S2 = 2 * H;
    const Lanes<G> ln(lane);
    const uint32_t g = ln.g;
    SplitLane<G, L> K;
    load_row<L>(K.n, A.mod.n, g);
    K.n0inv = A.mod.n0inv;
    K.rows_ = A.mod.rows;
    K.row_a = lds_row;
    K.row_c = lds_row + H;
    const int per = (1 << A.window) - 1;
    const uint64_t n_iter = (A.rows + total_slots - 1) / total_slots;
    for (uint64_t it = 0; it < n_iter; ++it) {
        uint64_t pos = slot + it * (uint64_t)total_slots;
        const bool live = pos < A.rows;
        if (!live) pos = A.rows - 1;
        const uint64_t r = A.order ? (uint64_t)A.order[pos] : pos;
        const uint64_t first = A.row_ptr ? A.row_ptr[r] : r * A.batch;
        const uint64_t count = A.row_ptr ? A.row_ptr[r + 1] - first : A.batch;
        uint32_t X0[L], X1[L], Y0[L], Y1[L];
        load_row<L>(X0, wave::reread_ptr(A.mod.e), g);
        load_row<L>(X1, wave::reread_ptr(A.mod.e) + H, g);
        for (int wi = A.n_windows - 1; wi >= 0; --wi) {
            if (wi != A.n_windows - 1)
                for (int s = 0; s < A.window; ++s) split_square<G, L>(X0, X1, K, ln);
            for (uint64_t el = 0; wave::ballot(el < count) != 0; ++el) {  // to the wavefront's longest row
                uint32_t d = 0;
                uint64_t tbl_item = 0;
                if (el < count) {
                    const uint64_t entry = first + el;
                    d = exp_digit(A.exps + entry * (uint64_t)A.exp_limbs, A.exp_limbs, wi * A.window, A.window);
                    const uint64_t col = A.cols ? (uint64_t)A.cols[entry] : el;
                    const uint64_t sg = (A.neg && A.neg[entry]) ? 1u : 0u;
                    tbl_item = col * (uint64_t)A.signs + sg;
                }
                if (wave::ballot(d != 0) != 0) {
                    const uint32_t* src = d ? A.table + (tbl_item * (uint64_t)per + (d - 1)) * S2 : A.mod.e;
                    load_row<L>(Y0, src, g);
                    load_row<L>(Y1, src + H, g);
                    split_mul<G, L>(X0, X1, Y0, Y1, K, ln);
                }
            }
        }
        split_exit<G, L>(A.out + r * (uint64_t)A.out_limbs, A.out_limbs, X0, X1, nullptr, 0, A.mod, K, ln, live);
    }
}

// ---- the tail of CRT decryption, one ciphertext per WAVEFRONT ----------------------------------------------------------
// decrypt_tail.h runs the O(h^2) tail (L-function, * hp, CRT) one ciphertext per thread: < 0.1 % of a large batch's work, but
// a serial chain of ~5 h^2 multiply-adds through LDS — 0.38 ms at 2048-bit keys, a fifth of the time of ONE decryption on a
// wave pair.  For small batches the same steps run here on the whole-wave geometry (G = 64, limbs of 29 bits, R = 2^(29 rows)),
// on the sweeps of this file and of mont_core.h:
//     L_p = (x_p - 1) / p          exact: ((x_p + R - 1) mod R) * p^-1 mod R, the low half of one plain product (mul_wide)
//     m_p = L_p * hp mod p         one Montgomery product against hp*R mod p, made canonical      (same for q)
//     u   = (m_q - m_p) * p^-1 mod q   with m_q - m_p + q formed as m_q + q + (R - 1 - m_p) + 1 - R: no borrow across lanes
//     m   = m_p + u * p            one plain product with m_p as its addend
// Reference: phe/paillier.py:346-354 (raw_decrypt), :362-364 (l_function), :366-374 (crt).  Same bits as decrypt_tail_one.
struct TailWaveConsts {
    const uint32_t *p, *q;          // H limbs each
    const uint32_t *pinv, *qinv;    // p^-1, q^-1 mod R
    const uint32_t *hp_r, *hq_r;    // hp * R mod p, hq * R mod q        (hp, hq: phe/paillier.py:234-235)
    const uint32_t* pinvq_r;        // p_inverse * R mod q               (p_inverse: phe/paillier.py:233)
    uint32_t p0inv, q0inv;          // -p^-1, -q^-1 mod 2^29
    int rows;                       // limbs of a number here (covers q with 4 spare bits, rows < H)
};
struct TailWaveArgs {
    TailWaveConsts k;
    const uint32_t* xp;  // (batch, x_stride) c^(p-1) mod p^2, canonical 32-bit words
    const uint32_t* xq;
    int x_stride;
    uint32_t* m_out;     // (batch, out_limbs)
    int out_limbs;
    uint64_t batch;
};

// lds: 2H + kLdsPad words of the wave
template <int L>
PHE_DEV void decrypt_tail_wave_body(const TailWaveArgs& A, uint32_t* lds, uint64_t item, uint32_t lane) {
    constexpr int G = 64, H = G * L;
    const Lanes<G> ln(lane);
    const uint32_t g = ln.g;
    const int rows = A.k.rows;
    uint32_t* row_a = lds;
    uint32_t* row_lo = lds + H;
    bool in[L];
#pragma unroll
    for (int k = 0; k < L; ++k) in[k] = (int)g * L + k < rows;
    uint32_t zero[L], mp[L], mq[L], np[L], nq[L];
#pragma unroll
    for (int k = 0; k < L; ++k) zero[k] = 0u;
    load_row<L>(np, A.k.p, g);
    load_row<L>(nq, A.k.q, g);
    // m_n = ((x - 1) / n) * h mod n for one CRT half
    const auto half = [&](uint32_t (&out)[L], const uint32_t* x, const uint32_t (&n)[L], uint32_t n0inv, const uint32_t* ninv_row,
                          const uint32_t* h_row) {
        uint32_t a[L], t[L], cst[L], hi[L];
        load_u32_as_r29<L>(a, x, A.x_stride, 0, g, rows);  // x mod R
        load_u32_as_r29<L>(t, x, A.x_stride, rows, g, rows);  // the rest of x: only to see whether x is zero
        uint32_t any = 0;
#pragma unroll
        for (int k = 0; k < L; ++k) any |= a[k] | t[k];
        // x = 0: a ciphertext that is a multiple of this prime (decrypt_tail.h tail_l_function): the reference's floor
        // division gives (0 - 1) // n = -1, i.e. n - 1 modulo n
        const bool x_is_zero = wave::ballot(any != 0u) == 0ull;
#pragma unroll
        for (int k = 0; k < L; ++k) t[k] = in[k] ? kLimbMask : 0u;
        add_normalize<G, L>(a, t, ln);  // + (R - 1): x - 1 modulo R ...
#pragma unroll
        for (int k = 0; k < L; ++k) a[k] = in[k] ? a[k] : 0u;  // ... the carry out of R dropped
        lds_put<L>(row_a, a, g);
        load_row<L>(cst, ninv_row, g);
        mul_wide<G, L>(hi, row_a, cst, zero, row_lo, ln, rows);  // row_lo: (x - 1) * n^-1 mod R = (x - 1) / n, canonical digits
        if (x_is_zero) {
#pragma unroll
            for (int k = 0; k < L; ++k) cst[k] = n[k] - ((g == 0u && k == 0) ? 1u : 0u);  // n is odd: no borrow
            lds_put<L>(row_lo, cst, g);
        }
        wave::lds_fence();
        load_row<L>(cst, h_row, g);
        montmul<G, L>(out, row_lo, cst, n, n0inv, ln, rows);
        canonicalize<G, L>(out, n, ln);
    };
    half(mp, A.xp + item * (uint64_t)A.x_stride, np, A.k.p0inv, A.k.pinv, A.k.hp_r);
    half(mq, A.xq + item * (uint64_t)A.x_stride, nq, A.k.q0inv, A.k.qinv, A.k.hq_r);
    // d = m_q - m_p + q  (0 < d < 2q)
    uint32_t d[L], t[L];
#pragma unroll
    for (int k = 0; k < L; ++k) {
        d[k] = mq[k];
        t[k] = in[k] ? (kLimbMask - mp[k]) : 0u;  // R - 1 - m_p
    }
    add_normalize<G, L>(d, nq, ln);
    add_normalize<G, L>(d, t, ln);
#pragma unroll
    for (int k = 0; k < L; ++k) t[k] = (g == 0u && k == 0) ? 1u : 0u;
    add_normalize<G, L>(d, t, ln);
    normalize_full<G, L>(d, ln);  // = d + R exactly: limb `rows` is 1
#pragma unroll
    for (int k = 0; k < L; ++k) d[k] = in[k] ? d[k] : 0u;
    // u = d * p^-1 mod q
    uint32_t u[L], cst[L];
    lds_put<L>(row_a, d, g);
    load_row<L>(cst, A.k.pinvq_r, g);
    montmul<G, L>(u, row_a, cst, nq, A.k.q0inv, ln, rows);
    canonicalize<G, L>(u, nq, ln);
    // m = m_p + u * p
    uint32_t hi[L], lo[L];
    lds_put<L>(row_a, u, g);
    mul_wide<G, L>(hi, row_a, np, mp, row_lo, ln, rows);
    wave::lds_fence();
    load_row<L>(lo, row_lo, g);
#pragma unroll
    for (int k = 0; k < L; ++k) lo[k] = in[k] ? lo[k] : 0u;
    normalize_full<G, L>(hi, ln);
    store_pair_as_u32<G, L>(A.m_out + item * (uint64_t)A.out_limbs, A.out_limbs, lo, hi, lds, g, true, rows);
}

}  // namespace phe
