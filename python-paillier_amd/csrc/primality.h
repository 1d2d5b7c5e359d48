// primality.h — batched Miller–Rabin with one modulus PER ROW (key generation).
//
// The reference finds its primes with getprimeover (phe/util.py:106-124): a random N-bit start with the top bit set, then
// gmpy2.next_prime / Crypto's getPrime / the pure-Python is_prime + miller_rabin of phe/util.py:381-443 — i.e. a
// run of strong-probable-prime tests  a^d = 1  or  a^(d 2^j) = -1 (mod n),  n - 1 = d 2^s,  over candidate after
// candidate.  Here every candidate (and, for the confirmation rounds, every base) is a row of one launch: a limb
// group holds its own modulus n and derives everything the Montgomery core (mont_core.h) needs on the device —
//   -n^-1 mod 2^29  (Newton),   R mod n  from  2^bits(n) - n  by modular doublings,   R^2 mod n  by a doubling ladder of
//   Montgomery squarings (mont(R 2^a, R 2^b) = R 2^(a+b))
// — so the host ships only candidates and bases.  The exponentiation is a plain left-to-right binary ladder over the
// 32*limbs bit positions (leading zeros square 1), with the multiply by the base always executed and selected per
// row, so that all groups of a wavefront stay converged; the last s squarings are the -1 checks.
// Written against mont_core.h / the wave:: primitives only: tests/emu compiles it for the host.
#pragma once
#include <stdint.h>

#include "mont_core.h"

namespace phe {

struct MillerRabinArgs {
    const uint32_t* n;     // (batch, limbs) little-endian 32-bit words: odd, > 3, bits(n) + 4 <= 29 * G * L
    const uint32_t* base;  // (batch, limbs): 2 <= base <= n - 2
    int limbs;
    uint8_t* pass;         // (batch): 1 = strong probable prime to this base
    uint64_t batch;
};

// all L limbs of every lane of the group equal?
template <int G, int L>
PHE_DEV bool group_equal(const uint32_t (&a)[L], const uint32_t (&b)[L], const Lanes<G>& ln) {
    uint32_t diff = 0;
#pragma unroll
    for (int k = 0; k < L; ++k) diff |= a[k] ^ b[k];
    const uint64_t ne = wave::ballot(diff != 0);
    const uint64_t grp = ((G == 64) ? ~0ull : ((1ull << G) - 1ull)) << (ln.lane - ln.g);
    return (ne & grp) == 0;
}

// x <- 2x mod n for canonical x < n
template <int G, int L>
PHE_DEV void double_mod(uint32_t (&x)[L], const uint32_t (&n)[L], const Lanes<G>& ln) {
    uint32_t t[L];
#pragma unroll
    for (int k = 0; k < L; ++k) t[k] = x[k];
    add_normalize<G, L>(x, t, ln);
    normalize_full<G, L>(x, ln);
    cond_sub<G, L>(x, n, ln);
}

template <int G, int L>
PHE_DEV void miller_rabin_body(const MillerRabinArgs& A, uint32_t* lds_row, uint32_t slot, uint32_t total_slots,
                               uint32_t lane) {
    constexpr int S = G * L;
    const Lanes<G> ln(lane);
    const uint32_t g = ln.g;
    const uint64_t n_iter = (A.batch + total_slots - 1) / total_slots;
    for (uint64_t it = 0; it < n_iter; ++it) {
        uint64_t item = slot + it * (uint64_t)total_slots;
        const bool live = item < A.batch;
        if (!live) item = A.batch - 1;
        const uint32_t* nw = A.n + item * (uint64_t)A.limbs;
        // ---- scalar facts about n, recomputed by every lane of the group: bit length, trailing zeros of n - 1 ----
        int bits = 0, tz = 0;
        bool seen = false;
        for (int w = 0; w < A.limbs; ++w) {
            uint32_t v = nw[w];
            if (v) bits = 32 * w + 32 - __builtin_clz(v);
            if (w == 0) v &= ~1u;  // n - 1
            if (!seen && v) {
                tz = 32 * w + __builtin_ctz(v);
                seen = true;
            }
        }
        uint32_t ninv = nw[0];  // n^-1 mod 2^32 (Newton: 3 -> 6 -> 12 -> 24 -> 48 bits)
        for (int i = 0; i < 5; ++i) ninv *= 2u - nw[0] * ninv;
        const uint32_t n0inv = (0u - ninv) & kLimbMask;
        uint32_t n[L], x[L], a[L], one[L], mone[L], t[L], u[L];
        load_u32_as_r29<L>(n, nw, A.limbs, 0, g);
        // ---- R mod n: 2^bits - n (< n because 2^(bits-1) <= n), then 29 S - bits modular doublings ----
#pragma unroll
        for (int k = 0; k < L; ++k) {
            const int lo = kRadixBits * ((int)g * L + k);  // bit position of this limb
            uint32_t keep = 0;
            if (bits > lo) keep = (bits - lo >= kRadixBits) ? kLimbMask : ((1u << (bits - lo)) - 1u);
            x[k] = (~n[k]) & keep;
        }
        if (g == 0u) x[0] += 1u;
        normalize_full<G, L>(x, ln);
        // (2^bits - n needs at most `bits` bits: the +1 cannot carry out of them because n > 0)
        // (the groups of a wavefront may hold moduli of different lengths: the trip count is the wave's maximum and
        // every group keeps its value once its own count is reached — double_mod contains ballots)
        const int doublings = kRadixBits * S - bits;
        for (int i = 0; wave::ballot(i < doublings) != 0; ++i) {
#pragma unroll
            for (int k = 0; k < L; ++k) t[k] = x[k];
            double_mod<G, L>(t, n, ln);
#pragma unroll
            for (int k = 0; k < L; ++k) x[k] = (i < doublings) ? t[k] : x[k];
        }
#pragma unroll
        for (int k = 0; k < L; ++k) one[k] = x[k];  // Montgomery form of 1, canonical
        // ---- n - (R mod n): Montgomery form of -1, canonical (R mod n is in [1, n) because n is odd > 1) ----
        {
            uint32_t br = 0, nz = 0;
#pragma unroll
            for (int k = 0; k < L; ++k) {
                const uint32_t v = n[k] - one[k] - br;
                br = v >> 31;
                mone[k] = v & kLimbMask;
                nz |= mone[k];
            }
            uint64_t out_top;
            const uint64_t bin = group_carry_in<G>(wave::ballot(br != 0), wave::ballot(nz == 0), out_top);
            uint32_t bi = lane_bit(bin, ln.lane);
#pragma unroll
            for (int k = 0; k < L; ++k) {
                const uint32_t v = mone[k] - bi;
                bi = v >> 31;
                mone[k] = v & kLimbMask;
            }
        }
        // ---- R^2 mod n = f(29 S) with f(k) = R 2^k mod n: f(2k) = mont(f(k), f(k)), f(k + 1) = 2 f(k) ----
        {
            const int E = kRadixBits * S;
            int top = 31 - __builtin_clz((uint32_t)E);
            double_mod<G, L>(x, n, ln);  // f(1)
            for (int b = top - 1; b >= 0; --b) {
                lds_put<L>(lds_row, x, g);
                montmul<G, L>(x, lds_row, x, n, n0inv, ln);
                canonicalize<G, L>(x, n, ln);
                if ((E >> b) & 1) double_mod<G, L>(x, n, ln);
            }
        }
        // ---- the base in Montgomery form, then a^((n-1) >> tz) by a binary ladder over all 32*limbs positions ----
        load_u32_as_r29<L>(t, A.base + item * (uint64_t)A.limbs, A.limbs, 0, g);
        lds_put<L>(lds_row, t, g);
        montmul<G, L>(a, lds_row, x, n, n0inv, ln);  // a * R^2 / R
#pragma unroll
        for (int k = 0; k < L; ++k) x[k] = one[k];
        for (int i = 32 * A.limbs - 1; i >= 0; --i) {
            lds_put<L>(lds_row, x, g);
            montmul<G, L>(t, lds_row, x, n, n0inv, ln);
            lds_put<L>(lds_row, t, g);
            montmul<G, L>(u, lds_row, a, n, n0inv, ln);
            const uint32_t bit = (i == 0) ? 0u : ((nw[i >> 5] >> (i & 31)) & 1u);  // bit i of n - 1
            const bool active = i >= tz;
#pragma unroll
            for (int k = 0; k < L; ++k) x[k] = active ? (bit ? u[k] : t[k]) : x[k];
        }
        // ---- x = a^d: pass if 1 or -1, else square up to tz - 1 times looking for -1 ----
#pragma unroll
        for (int k = 0; k < L; ++k) t[k] = x[k];
        canonicalize<G, L>(t, n, ln);
        const bool is_one = group_equal<G, L>(t, one, ln);  // both compares unconditionally: they contain ballots
        const bool is_mone = group_equal<G, L>(t, mone, ln);
        bool ok = is_one | is_mone;
        for (int j = 1; wave::ballot(j < tz && !ok) != 0; ++j) {
            lds_put<L>(lds_row, x, g);
            montmul<G, L>(x, lds_row, x, n, n0inv, ln);
#pragma unroll
            for (int k = 0; k < L; ++k) t[k] = x[k];
            canonicalize<G, L>(t, n, ln);
            const bool hit = group_equal<G, L>(t, mone, ln);
            if (j < tz && hit) ok = true;
        }
        if (live && g == 0u) A.pass[item] = ok ? 1 : 0;
    }
}

}  // namespace phe
