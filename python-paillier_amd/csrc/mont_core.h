// mont_core.h — fixed-width multi-precision Montgomery arithmetic for one 16-lane limb group.
//
// Replaces, for whole batches, what the reference delegates to gmpy2/libgmp one scalar at a time:
//   phe/util.py:38-50  powmod  -> modexp_uniform_body / modexp_var_body
//   phe/util.py:53-64  mulmod  -> mulmod_body
// and fuses the Paillier wrappers around them:
//   phe/paillier.py:102-139 raw_encrypt, :603-624 obfuscate, :328-354 raw_decrypt (the two CRT
//   half-exponentiations; the L/CRT tail is decrypt_tail.h), :705-719 _raw_add, :721-751 _raw_mul.
//
// Number layout.  A modulus N < W^S (W = 2^32, S = 16*L limbs) is handled by one DPP row of 16
// lanes; lane g keeps limbs [g*L, (g+1)*L) in VGPRs ("blocked" layout).  Four rows share a
// wavefront and never interact.  Montgomery radix is R = W^S.
//
// montmul() is word-serial CIOS: per limb a_i of the multiplier (broadcast-read from LDS, four
// limbs per ds_read_b128)
//     t += a_i * b            L  v_mad_u64_u32 + L+1 v_addc   (lane-local; carries between lanes
//     m  = t_0 * (-N^-1)      1  v_mul_lo + 1 DPP row_newbcast  are parked in a per-lane overflow
//     t += m * N              L  v_mad_u64_u32 + L+1 v_addc     word `th`, resolved once at the end)
//     t >>= 32                1  DPP row_shl:1 + 2 adds
// followed by one carry-lookahead across the row (two ballots + SALU mask arithmetic) and the
// final conditional subtraction, so every result is the canonical residue in [0, N) — which is
// what makes the GPU result bit-identical to gmpy2's.
//
// This header is written only against the `wave::` primitives (wave_gfx950.h on the device,
// tests/emu/wave_emu.h for CPU tests) and is otherwise plain C++.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace phe {

constexpr int kRow = 16;     // lanes per limb group
constexpr int kLdsPad = 4;   // words of padding after each row's LDS operand (keeps 16-B alignment)

constexpr uint64_t kRowLane0 = 0x0001000100010001ull;  // lane 0 of each of the 4 rows
constexpr uint64_t kRowTop = 0x8000800080008000ull;    // lane 15 of each row

PHE_DEV uint32_t lane_bit(uint64_t mask, uint32_t lane) { return (uint32_t)(mask >> lane) & 1u; }
PHE_DEV uint32_t row_top_bit(uint64_t mask, uint32_t lane) { return (uint32_t)(mask >> (lane | 15u)) & 1u; }

// Carry (or borrow) look-ahead across the 16 lanes of every row at once.
//   gen  : lanes whose lane-local add produced a carry out
//   prop : lanes that would pass an incoming carry on (all-ones sum / all-zero difference)
// gen and prop are disjoint by construction.  Returns the lanes that receive a carry-in;
// out_top gets (at the top-lane bit of each row) whether the row as a whole carried out.
PHE_DEV uint64_t row_carry_in(uint64_t gen, uint64_t prop, uint64_t& out_top) {
    const uint64_t gs = (gen << 1) & ~kRowLane0;
    const uint64_t pm = prop & ~kRowTop;  // the top lane must not ripple into the next row's field
    const uint64_t cin = (gs + pm) ^ pm;
    out_top = (gen | (prop & cin)) & kRowTop;
    return cin;
}

// Finish a lane-local addition: `c` is this lane's carry out of its L limbs.  Propagates carries
// lane to lane; returns (top-lane bits) the rows whose value overflowed W^S.
template <int L>
PHE_DEV uint64_t resolve_carries(uint32_t (&t)[L], uint32_t c, uint32_t lane) {
    uint32_t ones = 0xffffffffu;
#pragma unroll
    for (int k = 0; k < L; ++k) ones &= t[k];
    const uint64_t gen = wave::ballot(c != 0);
    const uint64_t prop = wave::ballot(ones == 0xffffffffu);
    uint64_t out_top;
    const uint64_t cin = row_carry_in(gen, prop, out_top);
    uint32_t ci = lane_bit(cin, lane);
#pragma unroll
    for (int k = 0; k < L; ++k) t[k] = wave::addc(t[k], 0u, ci, ci);
    return out_top;
}

// t <- t - N if t (plus the overflow bit ov_top) >= N.  Requires value < 2N.
template <int L>
PHE_DEV void cond_sub(uint32_t (&t)[L], const uint32_t (&n)[L], uint64_t ov_top, uint32_t lane) {
    uint32_t d[L];
    uint32_t bo = 0, nz = 0;
#pragma unroll
    for (int k = 0; k < L; ++k) {
        d[k] = wave::subb(t[k], n[k], bo, bo);
        nz |= d[k];
    }
    const uint64_t gen = wave::ballot(bo != 0);
    const uint64_t prop = wave::ballot(nz == 0);
    uint64_t out_top;
    const uint64_t bin = row_carry_in(gen, prop, out_top);
    const uint64_t take = ov_top | (~out_top & kRowTop);  // t >= N: overflowed, or no final borrow
    uint32_t bi = lane_bit(bin, lane);
    const uint32_t sel = row_top_bit(take, lane);
#pragma unroll
    for (int k = 0; k < L; ++k) {
        d[k] = wave::subb(d[k], 0u, bi, bi);
        t[k] = sel ? d[k] : t[k];
    }
}

// (a + b) mod N for a, b < N
template <int L>
PHE_DEV void modadd(uint32_t (&out)[L], const uint32_t (&a)[L], const uint32_t (&b)[L],
                    const uint32_t (&n)[L], uint32_t lane) {
    uint32_t t[L];
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < L; ++k) t[k] = wave::addc(a[k], b[k], c, c);
    const uint64_t ov = resolve_carries<L>(t, c, lane);
    cond_sub<L>(t, n, ov, lane);
#pragma unroll
    for (int k = 0; k < L; ++k) out[k] = t[k];
}

// t += 1 (value stays < W^S by the caller's guarantee)
template <int L>
PHE_DEV void add_one(uint32_t (&t)[L], uint32_t lane) {
    uint32_t c = ((lane & 15u) == 0u) ? 1u : 0u;
#pragma unroll
    for (int k = 0; k < L; ++k) t[k] = wave::addc(t[k], 0u, c, c);
    (void)resolve_carries<L>(t, c, lane);
}

// One CIOS row: t = (t + a_i*b + m*N) / W with the lane-local overflow kept in th (thh is the
// transient second overflow word; it is 0 on entry and exit).
template <int L>
PHE_DEV void mont_row(uint32_t (&t)[L], uint32_t& th, uint32_t ai, const uint32_t (&b)[L],
                      const uint32_t (&n)[L], uint32_t n0inv) {
    uint64_t p[L];
    uint32_t c, thh;
    // t += a_i * b
#pragma unroll
    for (int k = 0; k < L; ++k) p[k] = wave::mad(ai, b[k], t[k]);
    // quotient digit from the row's least significant word (lane 0, limb 0)
    const uint32_t m = wave::row_bcast0((uint32_t)p[0] * n0inv);
    t[0] = (uint32_t)p[0];
    c = 0;
#pragma unroll
    for (int k = 1; k < L; ++k) t[k] = wave::addc((uint32_t)p[k], (uint32_t)(p[k - 1] >> 32), c, c);
    th = wave::addc(th, (uint32_t)(p[L - 1] >> 32), c, c);
    thh = c;
    // t += m * N, written one limb down (the /W of this row)
#pragma unroll
    for (int k = 0; k < L; ++k) p[k] = wave::mad(m, n[k], t[k]);
    const uint32_t w0 = (uint32_t)p[0];  // lane 0: zero by construction; lane g>0: goes to lane g-1
    c = 0;
#pragma unroll
    for (int k = 1; k < L; ++k) t[k - 1] = wave::addc((uint32_t)p[k], (uint32_t)(p[k - 1] >> 32), c, c);
    const uint32_t top = wave::addc(th, (uint32_t)(p[L - 1] >> 32), c, c);
    thh += c;
    const uint32_t recv = wave::row_down1(w0);
    t[L - 1] = wave::addc(top, recv, 0u, c);
    th = thh + c;
}

// out = a * b * R^-1 mod N.   a: the row's S-limb multiplier in LDS (standard little-endian
// order), any value < R;  b < N in registers;  out in [0, N).
template <int L>
PHE_DEV void montmul(uint32_t (&out)[L], const uint32_t* a, const uint32_t (&b)[L],
                     const uint32_t (&n)[L], uint32_t n0inv, uint32_t lane) {
    constexpr int S = kRow * L;
    uint32_t t[L];
    uint32_t th = 0;
#pragma unroll
    for (int k = 0; k < L; ++k) t[k] = 0;
#pragma unroll 1
    for (int i = 0; i < S; i += 4) {
        const uint32_t a0 = a[i], a1 = a[i + 1], a2 = a[i + 2], a3 = a[i + 3];
        mont_row<L>(t, th, a0, b, n, n0inv);
        mont_row<L>(t, th, a1, b, n, n0inv);
        mont_row<L>(t, th, a2, b, n, n0inv);
        mont_row<L>(t, th, a3, b, n, n0inv);
    }
    // hand each lane's overflow word to the next lane, then resolve
    const uint32_t inc = wave::row_up1(th);
    uint32_t c;
    t[0] = wave::addc(t[0], inc, 0u, c);
#pragma unroll
    for (int k = 1; k < L; ++k) t[k] = wave::addc(t[k], 0u, c, c);
    uint64_t ov = resolve_carries<L>(t, c, lane);
    ov |= wave::ballot(th != 0) & kRowTop;  // the top lane's own overflow word is bit 32*S
    cond_sub<L>(t, n, ov, lane);
#pragma unroll
    for (int k = 0; k < L; ++k) out[k] = t[k];
}

// ---- operand movement ----------------------------------------------------------------------
// lane's L limbs of a `limbs`-word little-endian number at p, zero-extended to S words
template <int L>
PHE_DEV void load_limbs(uint32_t (&x)[L], const uint32_t* p, int limbs, uint32_t g) {
#pragma unroll
    for (int k = 0; k < L; ++k) {
        const int idx = (int)g * L + k;
        x[k] = (idx < limbs) ? p[idx] : 0u;
    }
}
template <int L>
PHE_DEV void store_limbs(uint32_t* p, const uint32_t (&x)[L], int limbs, uint32_t g) {
#pragma unroll
    for (int k = 0; k < L; ++k) {
        const int idx = (int)g * L + k;
        if (idx < limbs) p[idx] = x[k];
    }
}
// full-width (S words) row, e.g. window-table entries and per-modulus constants
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
// publish a value as the row's LDS multiplier operand
template <int L>
PHE_DEV void lds_put(uint32_t* row, const uint32_t (&x)[L], uint32_t g) {
    wave::lds_fence();
#pragma unroll
    for (int k = 0; k < L; ++k) row[g * L + k] = x[k];
    wave::lds_fence();
}

// ---- per-modulus constants (device pointers, S = 16*L words each) ---------------------------
struct ModConsts {
    const uint32_t* n;    // N
    const uint32_t* r1;   // R   mod N  (Montgomery one)
    const uint32_t* r2;   // R^2 mod N
    const uint32_t* r3;   // R^3 mod N  (folds the high half of a 2S-word input)
    const uint32_t* aux;  // encrypt: n*R mod n^2, so montmul(m, aux) = n*m
    uint32_t n0inv;       // -N^-1 mod 2^32
};

enum : int { kModeEncrypt = 0, kModeObfuscate = 1, kModeHalfDecrypt = 2, kModePow = 3 };

// Batch-uniform exponent (encrypt/obfuscate: e = n;  decrypt halves: e = p-1, q-1).
// The host turns e into a sliding-window schedule; every op word is (squarings << 8) | (idx+1)
// where idx selects the odd power base^(2*idx+1) from the row's table (0 = no multiply).
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
    uint32_t* table;  // scratch: total_rows * tbl_entries * S words
    uint64_t batch;
};

template <int L, int MODE>
PHE_DEV void modexp_uniform_body(const UniformArgs& A, uint32_t* lds_row, uint32_t row_slot,
                                 uint32_t total_rows, uint32_t lane) {
    constexpr int S = kRow * L;
    const uint32_t g = lane & 15u;
    const uint32_t n0inv = A.mod.n0inv;
    uint32_t n[L];
    load_row<L>(n, A.mod.n, g);
    uint32_t* tbl = A.table + (size_t)row_slot * (size_t)A.tbl_entries * S;

    // the wave iterates while any of its rows has work; an idle row recomputes the last item
    // harmlessly (all 64 lanes stay converged through the DPP/ballot steps of montmul)
    const uint64_t n_iter = (A.batch + total_rows - 1) / total_rows;
    for (uint64_t it = 0; it < n_iter; ++it) {
        uint64_t item = row_slot + it * (uint64_t)total_rows;
        const bool live = item < A.batch;
        if (!live) item = A.batch - 1;
        uint32_t acc[L], tmp[L], cst[L];
        // ---- base -> Montgomery form -------------------------------------------------------
        const uint32_t* bp = A.base + item * (uint64_t)A.base_limbs;
        load_limbs<L>(tmp, bp, A.base_limbs < S ? A.base_limbs : S, g);
        lds_put<L>(lds_row, tmp, g);
        load_row<L>(cst, A.mod.r2, g);
        montmul<L>(acc, lds_row, cst, n, n0inv, lane);
        if (MODE == kModeHalfDecrypt) {
            // c = lo + hi*W^S  ->  c*R = lo*R + hi*R^2  (mod N)
            load_limbs<L>(tmp, bp + S, A.base_limbs - S, g);
            lds_put<L>(lds_row, tmp, g);
            load_row<L>(cst, A.mod.r3, g);
            montmul<L>(tmp, lds_row, cst, n, n0inv, lane);
            modadd<L>(acc, acc, tmp, n, lane);
        }
        // ---- odd powers base^1, base^3, ... in Montgomery form ------------------------------
        store_row<L>(tbl, acc, g);
        if (A.tbl_entries > 1) {
            lds_put<L>(lds_row, acc, g);
            montmul<L>(cst, lds_row, acc, n, n0inv, lane);  // base^2
            for (int j = 1; j < A.tbl_entries; ++j) {
                lds_put<L>(lds_row, acc, g);
                montmul<L>(acc, lds_row, cst, n, n0inv, lane);
                store_row<L>(tbl + (size_t)j * S, acc, g);
            }
        }
        // ---- left-to-right sliding window --------------------------------------------------
        load_row<L>(acc, tbl + (size_t)A.first_idx * S, g);
        for (int op = 0; op < A.n_ops; ++op) {
            const uint32_t w = A.sched[op];
            const int nsq = (int)(w >> 8);
            const int sel = (int)(w & 0xffu);
            for (int s = 0; s < nsq; ++s) {
                lds_put<L>(lds_row, acc, g);
                montmul<L>(acc, lds_row, acc, n, n0inv, lane);
            }
            if (sel) {
                lds_put<L>(lds_row, acc, g);
                load_row<L>(tmp, tbl + (size_t)(sel - 1) * S, g);
                montmul<L>(acc, lds_row, tmp, n, n0inv, lane);
            }
        }
        // ---- leave Montgomery form (fused with the op's final multiply) ---------------------
        if (MODE == kModeEncrypt) {
            // nude ciphertext 1 + n*m (phe/paillier.py:134; the :125-130 branch is value-identical)
            load_limbs<L>(tmp, A.post + item * (uint64_t)A.post_limbs, A.post_limbs, g);
            lds_put<L>(lds_row, tmp, g);
            load_row<L>(cst, A.mod.aux, g);
            montmul<L>(tmp, lds_row, cst, n, n0inv, lane);  // n*m mod n^2  (<= n^2 - n)
            add_one<L>(tmp, lane);
        } else if (MODE == kModeObfuscate) {
            load_limbs<L>(tmp, A.post + item * (uint64_t)A.post_limbs, A.post_limbs, g);
        } else {
#pragma unroll
            for (int k = 0; k < L; ++k) tmp[k] = (g == 0u && k == 0) ? 1u : 0u;
        }
        // acc = x*R, tmp = y (standard form)  ->  montmul = x*y mod N, standard form
        lds_put<L>(lds_row, tmp, g);
        montmul<L>(acc, lds_row, acc, n, n0inv, lane);
        if (live) store_limbs<L>(A.out + item * (uint64_t)A.out_limbs, acc, A.out_limbs, g);
    }
}

// Per-element exponents (phe/paillier.py:751 powmod(c, scalar, n^2); :749 with the inverted base).
// Fixed 2^w-ary windows over the batch-wide maximum bit length: every row runs the same
// schedule (no divergence between the four rows of a wave); the digit only picks the table entry.
struct VarArgs {
    ModConsts mod;
    const uint32_t* base;  // (batch, base_limbs), values < N
    int base_limbs;
    const uint32_t* exps;  // (batch, exp_limbs)
    int exp_limbs;
    int window;      // w in 1..5
    int n_windows;   // ceil(max_bits / w), >= 1
    uint32_t* out;
    int out_limbs;
    uint32_t* table;  // scratch: total_rows * 2^w * S words
    uint64_t batch;
};

PHE_DEV uint32_t exp_digit(const uint32_t* e, int exp_limbs, int bitpos, int w) {
    // bits [bitpos, bitpos+w) of e
    const int wi = bitpos >> 5, sh = bitpos & 31;
    uint64_t v = (wi < exp_limbs) ? e[wi] : 0u;
    if (wi + 1 < exp_limbs) v |= (uint64_t)e[wi + 1] << 32;
    return (uint32_t)(v >> sh) & ((1u << w) - 1u);
}

template <int L>
PHE_DEV void modexp_var_body(const VarArgs& A, uint32_t* lds_row, uint32_t row_slot,
                             uint32_t total_rows, uint32_t lane) {
    constexpr int S = kRow * L;
    const uint32_t g = lane & 15u;
    const uint32_t n0inv = A.mod.n0inv;
    const int tbl_entries = 1 << A.window;
    uint32_t n[L];
    load_row<L>(n, A.mod.n, g);
    uint32_t* tbl = A.table + (size_t)row_slot * (size_t)tbl_entries * S;
    const uint64_t n_iter = (A.batch + total_rows - 1) / total_rows;
    for (uint64_t it = 0; it < n_iter; ++it) {
        uint64_t item = row_slot + it * (uint64_t)total_rows;
        const bool live = item < A.batch;
        if (!live) item = A.batch - 1;
        uint32_t acc[L], tmp[L], xm[L];
        load_limbs<L>(tmp, A.base + item * (uint64_t)A.base_limbs, A.base_limbs, g);
        lds_put<L>(lds_row, tmp, g);
        load_row<L>(xm, A.mod.r2, g);
        montmul<L>(xm, lds_row, xm, n, n0inv, lane);  // base*R
        // table: base^0 .. base^(2^w - 1)
        load_row<L>(acc, A.mod.r1, g);
        store_row<L>(tbl, acc, g);
        store_row<L>(tbl + S, xm, g);
#pragma unroll
        for (int k = 0; k < L; ++k) acc[k] = xm[k];
        for (int j = 2; j < tbl_entries; ++j) {
            lds_put<L>(lds_row, acc, g);
            montmul<L>(acc, lds_row, xm, n, n0inv, lane);
            store_row<L>(tbl + (size_t)j * S, acc, g);
        }
        const uint32_t* e = A.exps + item * (uint64_t)A.exp_limbs;
        uint32_t d = exp_digit(e, A.exp_limbs, (A.n_windows - 1) * A.window, A.window);
        load_row<L>(acc, tbl + (size_t)d * S, g);
        for (int wi = A.n_windows - 2; wi >= 0; --wi) {
            for (int s = 0; s < A.window; ++s) {
                lds_put<L>(lds_row, acc, g);
                montmul<L>(acc, lds_row, acc, n, n0inv, lane);
            }
            d = exp_digit(e, A.exp_limbs, wi * A.window, A.window);
            if (wave::ballot(d != 0) != 0) {  // wave-uniform: skip when all four rows have a zero digit
                lds_put<L>(lds_row, acc, g);
                load_row<L>(tmp, tbl + (size_t)d * S, g);
                montmul<L>(acc, lds_row, tmp, n, n0inv, lane);
            }
        }
#pragma unroll
        for (int k = 0; k < L; ++k) tmp[k] = (g == 0u && k == 0) ? 1u : 0u;
        lds_put<L>(lds_row, tmp, g);
        montmul<L>(acc, lds_row, acc, n, n0inv, lane);
        if (live) store_limbs<L>(A.out + item * (uint64_t)A.out_limbs, acc, A.out_limbs, g);
    }
}

// out = a*b mod N  (phe/util.py:53-64 mulmod; phe/paillier.py:705-719 _raw_add).  a < R, b < N.
// Row strides are explicit so the same kernel walks the pair-product tree of batched inversion.
struct MulArgs {
    ModConsts mod;
    const uint32_t* a;
    const uint32_t* b;
    uint32_t* out;
    size_t a_stride, b_stride, out_stride;  // words between consecutive rows
    int limbs;                              // words per number
    uint64_t batch;
};

template <int L>
PHE_DEV void mulmod_body(const MulArgs& A, uint32_t* lds_row, uint32_t row_slot, uint32_t total_rows,
                         uint32_t lane) {
    const uint32_t g = lane & 15u;
    const uint32_t n0inv = A.mod.n0inv;
    uint32_t n[L], r2[L];
    load_row<L>(n, A.mod.n, g);
    load_row<L>(r2, A.mod.r2, g);
    const uint64_t n_iter = (A.batch + total_rows - 1) / total_rows;
    for (uint64_t it = 0; it < n_iter; ++it) {
        uint64_t item = row_slot + it * (uint64_t)total_rows;
        const bool live = item < A.batch;
        if (!live) item = A.batch - 1;
        uint32_t x[L], y[L];
        load_limbs<L>(x, A.a + item * A.a_stride, A.limbs, g);
        load_limbs<L>(y, A.b + item * A.b_stride, A.limbs, g);
        lds_put<L>(lds_row, x, g);
        montmul<L>(x, lds_row, y, n, n0inv, lane);   // a*b/R
        lds_put<L>(lds_row, x, g);
        montmul<L>(x, lds_row, r2, n, n0inv, lane);  // a*b
        if (live) store_limbs<L>(A.out + item * A.out_stride, x, A.limbs, g);
    }
}

}  // namespace phe
