// decrypt_tail.h — the O(s^2) tail of CRT decryption, one ciphertext per thread.
//
// Reference: phe/paillier.py:346-354 (raw_decrypt), :362-364 (l_function), :366-374 (crt).
// Input are x_p = c^(p-1) mod p^2 and x_q = c^(q-1) mod q^2 from the two half-exponentiation
// launches (mont_core.h, kModeHalfDecrypt).  Per ciphertext:
//     L_p = (x_p - 1) / p                      exact division -> multiply by p^-1 mod W^h, low half (x_p = 0: -1, tail_l_function)
//     m_p = L_p * hp mod p                     one h-limb Montgomery product against hp*W^h mod p
//     (same for q)
//     u   = (m_q - m_p) * p^-1 mod q           m_p < p < q, so one conditional +q fixes the sign
//     m   = m_p + u * p
// This is < 0.1 % of a decryption's multiply-adds (h = limbs of p; ~9 h^2 vs ~1.2e3 * 2(2h)^2), so
// it is kept simple: thread-private arrays live in LDS, word i of thread t at ws[i*stride + t]
// (bank-conflict free), constants come through the scalar cache.
#pragma once
#include <stdint.h>

namespace phe {

struct TailConsts {
    int h;                  // limbs of p and of q
    const uint32_t* p;      // h words each ...
    const uint32_t* q;
    const uint32_t* pinvw;  // p^-1 mod W^h
    const uint32_t* qinvw;  // q^-1 mod W^h
    const uint32_t* hp_r;   // hp * W^h mod p   (hp: phe/paillier.py:234)
    const uint32_t* hq_r;   // hq * W^h mod q   (hq: phe/paillier.py:235)
    const uint32_t* pinvq_r;  // p_inverse * W^h mod q  (p_inverse: phe/paillier.py:233)
    uint32_t p0inv;         // -p^-1 mod 2^32
    uint32_t q0inv;         // -q^-1 mod 2^32
};

struct TailArgs {
    TailConsts k;
    const uint32_t* xp;  // (batch, x_stride) c^(p-1) mod p^2, standard form
    const uint32_t* xq;  // (batch, x_stride)
    int x_stride;
    uint32_t* m_out;  // (batch, out_limbs)
    int out_limbs;
    uint64_t batch;
};

constexpr int tail_ws_words(int h) { return 5 * h + 2; }

struct TailWs {
    uint32_t* base;
    int stride;
    PHE_DEV uint32_t& operator()(int off, int i) const { return base[(off + i) * stride]; }
};

// out[0..h) = ((x - 1) mod W^h) * kinv mod W^h   (tmp: h words)
PHE_DEV void tail_exact_div(const TailWs& ws, int out, int tmp, const uint32_t* x, const uint32_t* kinv, int h) {
    uint32_t borrow = 1;  // subtract 1
    for (int i = 0; i < h; ++i) {
        const uint32_t v = x[i];
        ws(tmp, i) = v - borrow;
        borrow = (v < borrow) ? 1u : 0u;
        ws(out, i) = 0;
    }
    for (int i = 0; i < h; ++i) {
        const uint32_t a = ws(tmp, i);
        uint32_t c = 0;
        for (int j = 0; j < h - i; ++j) {
            const uint64_t s = (uint64_t)a * kinv[j] + ws(out, i + j) + c;
            ws(out, i + j) = (uint32_t)s;
            c = (uint32_t)(s >> 32);
        }
    }
}

// l_function (phe/paillier.py:362-364: (x - 1) // p) for x = c^(p-1) mod p^2.  A ciphertext coprime to p gives x = 1 (mod p)
// and the division is exact; one that is NOT (c = 0, a multiple of p: nothing a holder of the public key can produce without
// knowing a factor of n, but the reference returns a value for it) gives x = 0, and the reference's floor division gives -1:
// out = p - 1 then, which the Montgomery product by hp turns into -hp mod p like the reference's (-1 * hp) % p.
PHE_DEV void tail_l_function(const TailWs& ws, int out, int tmp, const uint32_t* x, int x_words, const uint32_t* p,
                             const uint32_t* pinv, int h) {
    uint32_t any = 0;
    for (int i = 0; i < x_words; ++i) any |= x[i];
    if (any == 0) {
        for (int i = 0; i < h; ++i) ws(out, i) = p[i] - (i == 0 ? 1u : 0u);  // p is odd: no borrow
        return;
    }
    tail_exact_div(ws, out, tmp, x, pinv, h);
}

// out = a * b * W^-h mod n   (acc: h+2 words; a in ws, b and n in global memory; a < W^h, b < n)
PHE_DEV void tail_montmul(const TailWs& ws, int out, int acc, int a, const uint32_t* b, const uint32_t* n,
                          uint32_t n0inv, int h) {
    for (int i = 0; i < h + 2; ++i) ws(acc, i) = 0;
    for (int i = 0; i < h; ++i) {
        const uint32_t ai = ws(a, i);
        uint32_t c = 0;
        for (int j = 0; j < h; ++j) {
            const uint64_t s = (uint64_t)ai * b[j] + ws(acc, j) + c;
            ws(acc, j) = (uint32_t)s;
            c = (uint32_t)(s >> 32);
        }
        uint64_t s = (uint64_t)ws(acc, h) + c;
        ws(acc, h) = (uint32_t)s;
        ws(acc, h + 1) += (uint32_t)(s >> 32);
        const uint32_t m = ws(acc, 0) * n0inv;
        s = (uint64_t)m * n[0] + ws(acc, 0);
        c = (uint32_t)(s >> 32);
        for (int j = 1; j < h; ++j) {
            s = (uint64_t)m * n[j] + ws(acc, j) + c;
            ws(acc, j - 1) = (uint32_t)s;
            c = (uint32_t)(s >> 32);
        }
        s = (uint64_t)ws(acc, h) + c;
        ws(acc, h - 1) = (uint32_t)s;
        ws(acc, h) = ws(acc, h + 1) + (uint32_t)(s >> 32);
        ws(acc, h + 1) = 0;
    }
    // acc < 2n: subtract n once if acc >= n
    uint32_t borrow = 0;
    for (int i = 0; i < h; ++i) {
        const uint64_t d = (uint64_t)ws(acc, i) - n[i] - borrow;
        ws(out, i) = (uint32_t)d;
        borrow = (uint32_t)(d >> 63);
    }
    const bool ge = (ws(acc, h) != 0) || (borrow == 0);
    if (!ge)
        for (int i = 0; i < h; ++i) ws(out, i) = ws(acc, i);
}

PHE_DEV void decrypt_tail_one(const TailArgs& A, const TailWs& ws, uint64_t item) {
    const int h = A.k.h;
    // workspace map: R = [0, 2h+2) (acc = R[0..h+2), tmpA = R[h+2..2h+2)), MP, MQ, U
    const int R = 0, ACC = 0, TA = h + 2, MP = 2 * h + 2, MQ = 3 * h + 2, U = 4 * h + 2;
    const uint32_t* xp = A.xp + item * (uint64_t)A.x_stride;
    const uint32_t* xq = A.xq + item * (uint64_t)A.x_stride;
    // m_p
    tail_l_function(ws, TA, ACC, xp, A.x_stride, A.k.p, A.k.pinvw, h);
    tail_montmul(ws, MP, ACC, TA, A.k.hp_r, A.k.p, A.k.p0inv, h);
    // m_q
    tail_l_function(ws, TA, ACC, xq, A.x_stride, A.k.q, A.k.qinvw, h);
    tail_montmul(ws, MQ, ACC, TA, A.k.hq_r, A.k.q, A.k.q0inv, h);
    // d = (m_q - m_p) mod q
    uint32_t borrow = 0;
    for (int i = 0; i < h; ++i) {
        const uint64_t d = (uint64_t)ws(MQ, i) - ws(MP, i) - borrow;
        ws(TA, i) = (uint32_t)d;
        borrow = (uint32_t)(d >> 63);
    }
    if (borrow) {
        uint32_t c = 0;
        for (int i = 0; i < h; ++i) {
            const uint64_t s = (uint64_t)ws(TA, i) + A.k.q[i] + c;
            ws(TA, i) = (uint32_t)s;
            c = (uint32_t)(s >> 32);
        }
    }
    // u = d * p^-1 mod q
    tail_montmul(ws, U, ACC, TA, A.k.pinvq_r, A.k.q, A.k.q0inv, h);
    // m = m_p + u*p
    for (int i = 0; i < h; ++i) {
        ws(R, i) = ws(MP, i);
        ws(R, h + i) = 0;
    }
    for (int i = 0; i < h; ++i) {
        const uint32_t ui = ws(U, i);
        uint32_t c = 0;
        for (int j = 0; j < h; ++j) {
            const uint64_t s = (uint64_t)ui * A.k.p[j] + ws(R, i + j) + c;
            ws(R, i + j) = (uint32_t)s;
            c = (uint32_t)(s >> 32);
        }
        // carry ripples into the still-clean upper words
        for (int j = i + h; c != 0 && j < 2 * h; ++j) {
            const uint64_t s = (uint64_t)ws(R, j) + c;
            ws(R, j) = (uint32_t)s;
            c = (uint32_t)(s >> 32);
        }
    }
    uint32_t* out = A.m_out + item * (uint64_t)A.out_limbs;
    for (int i = 0; i < A.out_limbs; ++i) out[i] = (i < 2 * h) ? ws(R, i) : 0u;
}

}  // namespace phe
