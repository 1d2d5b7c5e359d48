// key_setup.h — host-side, once-per-key derivation of the constants the kernels consume.
//
// The reference derives its per-key values in PaillierPublicKey.__init__ (phe/paillier.py:86-90:
// n, nsquare, max_int) and PaillierPrivateKey.__init__ (:217-235: p<q, psquare, qsquare,
// p_inverse, hp, hq).  The Montgomery-domain companions needed by mont_core.h / decrypt_tail.h
// (R mod N, R^2, R^3, -N^-1 mod 2^32, n*R mod n^2, p^-1 mod W^h, hp*W^h mod p, the sliding-window
// schedule of an exponent) are computed here with a deliberately tiny schoolbook bignum: this is
// cold code, a few hundred microseconds per key.
//
// Pure C++ (no HIP): also compiled into the CPU emulator used by tests/.
#pragma once
#include <stdlib.h>
#include <stdint.h>

#include <algorithm>
#include <stdexcept>
#include <vector>

namespace phe {
namespace host {

using Big = std::vector<uint32_t>;  // little-endian limbs, fixed length chosen by the caller

inline Big big_from(const uint32_t* p, int limbs, int width) {
    Big r((size_t)width, 0u);
    for (int i = 0; i < limbs && i < width; ++i) r[(size_t)i] = p[i];
    for (int i = width; i < limbs; ++i)
        if (p[i]) throw std::invalid_argument("value does not fit the limb width");
    return r;
}
inline int big_bits(const Big& a) {
    for (int i = (int)a.size() - 1; i >= 0; --i)
        if (a[(size_t)i]) return 32 * i + 32 - __builtin_clz(a[(size_t)i]);
    return 0;
}
inline bool big_is_zero(const Big& a) { return big_bits(a) == 0; }
inline int big_cmp(const Big& a, const Big& b) {  // same length
    for (int i = (int)a.size() - 1; i >= 0; --i)
        if (a[(size_t)i] != b[(size_t)i]) return a[(size_t)i] < b[(size_t)i] ? -1 : 1;
    return 0;
}
inline uint32_t big_sub_inplace(Big& a, const Big& b) {  // a -= b, returns borrow
    uint64_t br = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        const uint64_t d = (uint64_t)a[i] - b[i] - br;
        a[i] = (uint32_t)d;
        br = (d >> 63) & 1u;
    }
    return (uint32_t)br;
}
inline uint32_t big_add_inplace(Big& a, const Big& b) {
    uint64_t c = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        const uint64_t s = (uint64_t)a[i] + b[i] + c;
        a[i] = (uint32_t)s;
        c = s >> 32;
    }
    return (uint32_t)c;
}
// a = 2a mod n  (a < n)
inline void big_double_mod(Big& a, const Big& n) {
    uint32_t top = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        const uint32_t v = a[i];
        a[i] = (v << 1) | top;
        top = v >> 31;
    }
    if (top || big_cmp(a, n) >= 0) big_sub_inplace(a, n);
}
// a * 2^k mod n
inline Big big_shift_mod(Big a, int k, const Big& n) {
    for (int i = 0; i < k; ++i) big_double_mod(a, n);
    return a;
}
inline Big big_mul(const Big& a, const Big& b) {  // full product, a.size()+b.size() limbs
    Big r(a.size() + b.size(), 0u);
    for (size_t i = 0; i < a.size(); ++i) {
        uint64_t c = 0;
        for (size_t j = 0; j < b.size(); ++j) {
            const uint64_t s = (uint64_t)a[i] * b[j] + r[i + j] + c;
            r[i + j] = (uint32_t)s;
            c = s >> 32;
        }
        r[i + b.size()] = (uint32_t)c;
    }
    return r;
}
inline Big big_resize(Big a, int width) {
    for (size_t i = (size_t)width; i < a.size(); ++i)
        if (a[i]) throw std::invalid_argument("value does not fit the limb width");
    a.resize((size_t)width, 0u);
    return a;
}
// a mod n for a < n * 2^(32*extra) by binary long division (cold path, sizes are tiny)
inline Big big_mod(const Big& a, const Big& n) {
    const int w = (int)n.size();
    Big r((size_t)w, 0u);
    for (int bit = big_bits(a) - 1; bit >= 0; --bit) {
        big_double_mod(r, n);
        if ((a[(size_t)(bit >> 5)] >> (bit & 31)) & 1u) {
            Big one((size_t)w, 0u);
            one[0] = 1;
            if (big_add_inplace(r, one) || big_cmp(r, n) >= 0) big_sub_inplace(r, n);
        }
    }
    return r;
}
// -n^-1 mod 2^32 for odd n
inline uint32_t neg_inv32(uint32_t n0) {
    uint32_t x = n0;  // correct to 3 bits
    for (int i = 0; i < 5; ++i) x *= 2u - n0 * x;
    return 0u - x;
}
// a^-1 mod 2^(32*h) for odd a (Newton/Hensel lifting on whole words)
inline Big inv_mod_pow2(const Big& a, int h) {
    Big x((size_t)h, 0u);
    x[0] = 0u - neg_inv32(a[0]);
    for (int have = 1; have < h; have *= 2) {
        // x <- x * (2 - a*x)  mod W^h
        Big ax = big_mul(a, x);
        ax.resize((size_t)h);
        Big two((size_t)h, 0u);
        two[0] = 2;
        big_sub_inplace(two, ax);
        Big nx = big_mul(x, two);
        nx.resize((size_t)h);
        x = nx;
    }
    return x;
}

// a = q*n + r by binary long division (cold path); q has a.size() limbs, r has n.size() limbs
inline void big_divmod(const Big& a, const Big& n, Big& q, Big& r) {
    const size_t w = n.size();
    Big rem(w + 1, 0u), nn(n);
    nn.push_back(0u);
    q.assign(a.size(), 0u);
    for (int bit = big_bits(a) - 1; bit >= 0; --bit) {
        uint32_t top = (a[(size_t)(bit >> 5)] >> (bit & 31)) & 1u;
        for (size_t i = 0; i <= w; ++i) {
            const uint32_t v = rem[i];
            rem[i] = (v << 1) | top;
            top = v >> 31;
        }
        if (big_cmp(rem, nn) >= 0) {
            big_sub_inplace(rem, nn);
            q[(size_t)(bit >> 5)] |= 1u << (bit & 31);
        }
    }
    rem.resize(w);
    r = rem;
}
// ---- radix-2^29 geometry -----------------------------------------------------------------------
constexpr int kRadixBits = 29;

// a Big (32-bit limbs) -> `count` limbs of 29 bits
inline std::vector<uint32_t> to_r29(const Big& a, int count) {
    std::vector<uint32_t> r((size_t)count, 0u);
    for (int j = 0; j < count; ++j) {
        const int bit = kRadixBits * j;
        const size_t q = (size_t)(bit >> 5);
        const int o = bit & 31;
        uint64_t w = q < a.size() ? a[q] : 0u;
        if (q + 1 < a.size()) w |= (uint64_t)a[q + 1] << 32;
        r[(size_t)j] = (uint32_t)(w >> o) & ((1u << kRadixBits) - 1u);
    }
    if (big_bits(a) > kRadixBits * count) throw std::invalid_argument("value does not fit the limb group");
    return r;
}

// Limb-group geometry: G lanes x L limbs of 29 bits, S = G*L.  The compiled kernels exist for the
// (G, L) pairs below; a modulus needs 29*S >= bits(N) + 4 (R = 2^(29 S) >= 16 N keeps products of
// values in [0, 2N) inside [0, 2N)), and S must also cover the caller's 32-bit-word width.
struct Geometry {
    int G = 0, L = 0;
    int S() const { return G * L; }
};
static const int kL16[] = {1, 2, 3, 5, 7, 9, 14, 18};
static const int kL8[] = {5, 9, 14, 18, 27};
static const int kL4[] = {9, 18, 27, 36};
static const int kL2[] = {18, 36};

// prefer_group: 0 = automatic (least padding, then fewest lanes = most limbs per lane, which amortises the
// per-digit DPP/quotient work best); 2 / 4 / 8 / 16 = the narrowest group allowed (wider ones are the fallback).
// Limbs per lane up to this many keep every 64-bit column accumulator below 2^64 for ANY operands:
// an accumulator stays in a lane for L digits and takes two products < 2^58 (+ almost-normalisation slack) per
// digit, so 2L * 2^58 < 2^64 needs L <= 31.  Larger L is only used after accumulators_fit() has checked the
// actual modulus.
constexpr int kMaxUnconditionalL = 31;

inline Geometry pick_geometry(int modulus_bits, int min_bits, int prefer_group, int max_L = 1 << 30) {
    const int need_bits = std::max(modulus_bits + 4, min_bits);
    const int need = (need_bits + kRadixBits - 1) / kRadixBits;
    Geometry best;
    auto consider = [&](int G, int L) {
        const int S = G * L;
        if (S < need || L > max_L) return;
        if (best.G == 0 || S < best.S() || (S == best.S() && L > best.L)) {
            best.G = G;
            best.L = L;
        }
    };
    if (prefer_group != 16) {
        const int narrowest = prefer_group == 0 ? 2 : prefer_group;
        if (narrowest <= 2)
            for (int L : kL2) consider(2, L);
        if (narrowest <= 4)
            for (int L : kL4) consider(4, L);
        if (narrowest <= 8)
            for (int L : kL8) consider(8, L);
    }
    // 16-lane groups: the only choice in latency mode, a competitor in automatic mode, the fallback otherwise
    if (prefer_group == 16 || prefer_group == 0 || best.G == 0)
        for (int L : kL16) consider(16, L);
    return best;  // G == 0: too wide for the compiled kernels
}

// Worst-case value of a column accumulator of mont_core.h:montmul for the modulus limbs `n29` split into lanes
// of L limbs: it enters a lane below 2^29, takes a_i*b[k] (both digits almost-normalised, < 2^29 + 2^8) and
// m*n[k] (m < 2^29) at each of the lane's L positions, and once the carry of the limb below (< 2^35).
inline bool accumulators_fit(const std::vector<uint32_t>& n29, int L) {
    const unsigned __int128 limit = (unsigned __int128)1 << 64;
    const unsigned __int128 digit = ((unsigned __int128)1 << 29) + 256;
    // entry value + carry of the limb below + the carry added by the final normalisation sweep, with margin
    const unsigned __int128 fixed = ((unsigned __int128)1 << 29) + ((unsigned __int128)1 << 37) + (unsigned __int128)L * digit * digit;
    for (size_t lane = 0; lane + (size_t)L <= n29.size(); lane += (size_t)L) {
        unsigned __int128 sum = 0;
        for (int k = 0; k < L; ++k) sum += n29[lane + (size_t)k];
        if (fixed + sum * (((unsigned __int128)1 << 29) - 1) >= limit) return false;
    }
    return true;
}

// Everything mont_core.h needs for one modulus: S limbs of 29 bits each, R = 2^(29 S).
struct ModulusPack {
    int G = 0, L = 0, S = 0;
    int bits = 0;
    std::vector<uint32_t> n, r1, r2, r3, aux;  // 29-bit limbs
    uint32_t n0inv = 0;                        // -N^-1 mod 2^29
};

// aux_src (optional, < N): aux = aux_src * R mod N.  min_bits: width the group must also cover
// (the caller's 32-bit-word rows; both CRT halves of a private key share one geometry).
inline ModulusPack build_modulus(const Big& N_any, const Big* aux_src, int min_bits = 0, int prefer_group = 0) {
    ModulusPack m;
    m.bits = big_bits(N_any);
    const int w32 = (m.bits + 31) / 32;
    Geometry geo = pick_geometry(m.bits, min_bits, prefer_group);
    if (geo.G == 0) throw std::invalid_argument("modulus too wide for the compiled kernels (max 8344 bits)");
    if (geo.L > kMaxUnconditionalL && !accumulators_fit(to_r29(big_resize(N_any, w32), geo.S()), geo.L)) {
        // this modulus has a lane of unusually large limbs (probability ~1e-9 for a random one): stay at L <= 31
        geo = pick_geometry(m.bits, min_bits, prefer_group, kMaxUnconditionalL);
        if (geo.G == 0) throw std::invalid_argument("modulus too wide for the compiled kernels (max 8344 bits)");
    }
    m.G = geo.G;
    m.L = geo.L;
    m.S = geo.S();
    const Big N = big_resize(N_any, w32);
    if ((N[0] & 1u) == 0u) throw std::invalid_argument("modulus must be odd");
    Big one((size_t)w32, 0u);
    one[0] = 1;
    if (big_cmp(one, N) >= 0) throw std::invalid_argument("modulus must be > 1");
    const int rbits = kRadixBits * m.S;
    const Big r1 = big_shift_mod(one, rbits, N);
    const Big r2 = big_shift_mod(r1, rbits, N);
    const Big r3 = big_shift_mod(r2, rbits, N);
    m.n = to_r29(N, m.S);
    m.r1 = to_r29(r1, m.S);
    m.r2 = to_r29(r2, m.S);
    m.r3 = to_r29(r3, m.S);
    if (aux_src) {
        Big a = big_resize(*aux_src, w32);
        if (big_cmp(a, N) >= 0) throw std::invalid_argument("aux value must be < modulus");
        m.aux = to_r29(big_shift_mod(a, rbits, N), m.S);
    } else {
        m.aux.assign((size_t)m.S, 0u);
    }
    m.n0inv = neg_inv32(N[0]) & ((1u << kRadixBits) - 1u);
    return m;
}

// ---- split-modulus ("n-adic") arithmetic: constants of csrc/split_core.h -----------------------------------
// Work modulo n^2 is done on pairs (X0, X1), x*R = X0 - n*X1 (mod n^2), with half-width Montgomery passes modulo n
// only: R = 2^(29 H) >= 16 n, H = G*L limbs.  A fused sweep adds three products per digit to a column accumulator
// (3L * 2^58 < 2^64: L <= 21); wider lanes (L = 27) run the words of a pair product as single sweeps of two.
static const int kS16[] = {1, 2, 3, 4, 5, 7, 9, 14, 18};
static const int kS8[] = {3, 5, 7, 9, 14, 18};
static const int kS4[] = {5, 9, 14, 18, 27};
static const int kS2[] = {9, 18, 27};
// ONE lane per number (G = 1): no cross-lane step at all — the natural throughput geometry of the CRT halves of 1024-bit keys
// (p, q of 512 bits = 18 limbs).  Asked for with prefer_group = 1 (the private side's rung 0 only).
static const int kS1[] = {18};
constexpr int kMaxSplitL = 31;  // two products per digit per sweep: 2L * 2^58 < 2^64 (fused sweeps, three products: L <= 21)
// G = 64: ONE number per wavefront (wave_gfx950.h: wave-wide DPP shifts, scalar broadcast) — the latency rung for a handful
// of numbers.  Its numbers need not fill the 64*L limbs: SplitPack::rows (a multiple of L) is what a sweep runs over.
static const int kS64[] = {1, 2, 3, 5};

inline Geometry pick_geometry_split(int n_bits, int prefer_group) {
    const int need = (n_bits + 4 + kRadixBits - 1) / kRadixBits;
    Geometry best;
    if (prefer_group == 64) {
        for (int L : kS64)
            if (64 * L >= need) {
                best.G = 64;
                best.L = L;
                return best;
            }
        return best;  // wider than 64 x 5 limbs (keys above ~9200 bits): no whole-wave kernel
    }
    auto consider = [&](int G, int L) {
        const int S = G * L;
        if (S < need) return;
        if (best.G == 0 || S < best.S() || (S == best.S() && L > best.L)) {
            best.G = G;
            best.L = L;
        }
    };
    if (prefer_group != 16) {
        const int narrowest = prefer_group == 0 ? 2 : prefer_group;
        if (narrowest <= 1)
            for (int L : kS1) consider(1, L);
        if (narrowest <= 2)
            for (int L : kS2) consider(2, L);
        if (narrowest <= 4)
            for (int L : kS4) consider(4, L);
        if (narrowest <= 8)
            for (int L : kS8) consider(8, L);
    }
    if (prefer_group == 16 || prefer_group == 0 || best.G == 0)
        for (int L : kS16) consider(16, L);
    return best;  // G == 0: no compiled split kernel covers this width
}

struct SplitPack {
    int G = 0, L = 0, H = 0;  // H = G*L limbs of 29 bits cover n (+4 bits)
    int rows = 0;             // limbs a number really has, R = 2^(29 rows): H, except for G = 64 (the least multiple of L that covers n)
    int bits = 0;             // bits of n
    int chunks = 0;           // conv rows available: inputs of up to chunks*29*H bits
    std::vector<uint32_t> n, r1;  // H limbs each: n, R mod n
    std::vector<uint32_t> e;      // pair of 1:       R mod n^2       = E0 - n*E1      (E0 | E1, 2H)
    std::vector<uint32_t> conv;   // chunk j:         R^(j+2) mod n^2 = D0 - n*D1      (chunks * 2H)
    std::vector<uint32_t> nsq;    // n^2, 2H limbs
    uint32_t n0inv = 0;           // -n^-1 mod 2^29
};

// max_input_bits: widest number that will be converted into the pair representation (a ciphertext).
// whole_L / whole_rows (both or none): a pack with exactly that lane width and that many limbs per number on groups of whole_G
// lanes (the whole wave, or 16-lane groups: the geometries whose row count is a run-time value, split_core.h SplitLane::rows)
inline SplitPack build_split(const Big& n_any, int max_input_bits, int prefer_group = 0, int whole_L = 0, int whole_rows = 0,
                             int whole_G = 64) {
    SplitPack P;
    P.bits = big_bits(n_any);
    const int w = (P.bits + 31) / 32;
    Geometry geo = pick_geometry_split(P.bits, prefer_group);
    if (whole_L) {
        geo.G = whole_G;
        geo.L = whole_L;
    }
    if (geo.G == 0) return P;  // caller falls back to the full-width kernels
    P.G = geo.G;
    P.L = geo.L;
    P.H = geo.S();
    P.rows = P.H;
    if (whole_L) {
        if (whole_rows % whole_L || whole_rows > P.H || kRadixBits * whole_rows < P.bits + 4) throw std::invalid_argument("bad row count for the pack");
        P.rows = whole_rows;
    } else if (P.G == 64) {
        // a multiple of the digits a whole-wave sweep takes per trip (split_core.h Trip<64, L>: 4, 4, 6, 10)
        const int need = (P.bits + 4 + kRadixBits - 1) / kRadixBits;
        const int trip = (P.L == 1 ? 4 : 2) * P.L;
        P.rows = (need + trip - 1) / trip * trip;
        if (P.rows > P.H) P.rows = P.H;  // (64*L is a multiple of every trip length)
    }
    const Big n = big_resize(n_any, w);
    if ((n[0] & 1u) == 0u) throw std::invalid_argument("modulus must be odd");
    const int rbits = kRadixBits * P.rows;
    P.chunks = std::max(1, (max_input_bits + rbits - 1) / rbits);
    const Big nsq = big_resize(big_mul(n, n), 2 * w);
    Big one((size_t)w, 0u);
    one[0] = 1;
    P.n = to_r29(n, P.H);
    P.r1 = to_r29(big_shift_mod(one, rbits, n), P.H);
    {   // n^2 = lo + hi * R as two rows of H limbs (for rows == H simply its 2H limbs)
        const std::vector<uint32_t> all = to_r29(nsq, 2 * P.rows);
        P.nsq.assign((size_t)(2 * P.H), 0u);
        for (int k = 0; k < P.rows; ++k) {
            P.nsq[(size_t)k] = all[(size_t)k];
            P.nsq[(size_t)(P.H + k)] = all[(size_t)(P.rows + k)];
        }
    }
    P.n0inv = neg_inv32(n[0]) & ((1u << kRadixBits) - 1u);
    // Z = z0 + z1*n (Z < n^2)  ->  the pair (z0, -z1 mod n)
    auto pair_of = [&](const Big& Z, std::vector<uint32_t>& out) {
        Big q, r;
        big_divmod(Z, n, q, r);
        Big z1 = big_resize(q, w), neg = n;
        big_sub_inplace(neg, z1);
        if (big_is_zero(z1)) neg = z1;
        const std::vector<uint32_t> a = to_r29(r, P.H), b = to_r29(neg, P.H);
        out.insert(out.end(), a.begin(), a.end());
        out.insert(out.end(), b.begin(), b.end());
    };
    Big one2((size_t)(2 * w), 0u);
    one2[0] = 1;
    Big Z = big_shift_mod(one2, rbits, nsq);  // R mod n^2
    pair_of(Z, P.e);
    for (int j = 0; j < P.chunks; ++j) {
        Z = big_shift_mod(Z, rbits, nsq);  // R^(j+2) mod n^2
        pair_of(Z, P.conv);
    }
    return P;
}

// One number on a PAIR of wavefronts (split_core.h modexp_split_ab_body): the sweeps run modulo the scaled modulus
// n~ = k*n, k = -n^-1 mod 2^29, so that n~ = -1 (mod 2^29) and a quotient digit is the accumulator's low digit as it is; the
// pair form inside is the one modulo n~^2 (n^2 divides it), and the way out works modulo the true n with the same
// R = 2^(29 rows): X0 - n~*X1 = X0 - n*(k*X1), so split_exit multiplies X1 by k*(n-1) where it multiplies by n-1 otherwise.
// rows covers n~ with the 4 bits of slack the lazy bounds need: one limb more than n's own, no rounding to a trip length.
// Round 4: the same constants serve the single-wave kernels on the late sweeps (split_core.h pair_late / modexp_split_late_body)
// on the whole wave AND on 16-lane groups (G = 16: the rung's own lane width, rows = the least multiple of it that covers n~).
struct QuickPack {
    SplitPack scaled;            // constants modulo n~ (G = 64 or 16)
    SplitPack exit;              // constants modulo n, same G, L and rows
    std::vector<uint32_t> nbar;  // (n~ + 1) / 2^29, H limbs
    std::vector<uint32_t> kx;    // k*(n - 1) mod n, H limbs
    bool ok() const { return scaled.G != 0; }
};
constexpr int kMaxLateL = 9;  // lane widths the late single-wave kernels are instantiated for (split_kernels.inc)

// lane width a QuickPack of this modulus needs (0: wider than the whole-wave kernels go)
inline int quick_lane_width(const Big& n_any) {
    const int need = (big_bits(n_any) + kRadixBits + 4 + kRadixBits - 1) / kRadixBits;
    for (int L : kS64)
        if (64 * L > (need + L - 1) / L * L) return L;
    return 0;
}

inline QuickPack build_quick(const Big& n_any, int max_input_bits, int L, int G = 64) {
    QuickPack Q;
    if (L == 0) return Q;
    const int bits = big_bits(n_any), w = (bits + 31) / 32;
    const Big n = big_resize(n_any, w);
    if ((n[0] & 1u) == 0u) throw std::invalid_argument("modulus must be odd");
    const uint32_t k = neg_inv32(n[0]) & ((1u << kRadixBits) - 1u);
    Big nk = big_mul(n, Big(1, k));
    const int kbits = big_bits(nk);
    nk = big_resize(nk, (kbits + 31) / 32);
    const int need = (kbits + 4 + kRadixBits - 1) / kRadixBits;
    const int rows = (need + L - 1) / L * L;
    // the whole wave: a sweep of the wave-pair kernels is at least two trips (split_core.h AbTrip), and word `rows` of a digit row
    // is used (the quotient row has rows + 1 words); 16-lane groups (late single-wave kernels only): the lanes must hold n~
    if (G == 64 ? (rows + 1 > 64 * L || rows < (L == 1 ? 4 : 2) * L) : (G != 16 || rows > 16 * L || L > kMaxLateL)) return Q;
    Q.scaled = build_split(nk, max_input_bits, G, L, rows, G);
    Q.exit = build_split(n, max_input_bits, G, L, rows, G);
    if (Q.scaled.n0inv != 1u) throw std::logic_error("scaled modulus is not -1 modulo the radix");
    {   // (n~ + 1) / 2^29: limb 0 of n~ + 1 is zero
        Big one(nk.size() + 1, 0u), up = big_resize(nk, (int)nk.size() + 1);
        one[0] = 1;
        big_add_inplace(up, one);
        const std::vector<uint32_t> all = to_r29(up, G * L + 1);
        if (all[0] != 0u) throw std::logic_error("scaled modulus + 1 is not a multiple of the radix");
        Q.nbar.assign(all.begin() + 1, all.end());
    }
    {   // k*(n - 1) mod n
        Big one((size_t)w, 0u), nm1 = n, quo, rem;
        one[0] = 1;
        big_sub_inplace(nm1, one);
        big_divmod(big_mul(nm1, Big(1, k)), n, quo, rem);
        Q.kx = to_r29(big_resize(rem, w), G * L);
    }
    return Q;
}

// Left-to-right sliding-window schedule for a batch-uniform exponent e > 0.
// Table entry idx holds base^(2*idx+1).  Op word = (squarings << 8) | (idx + 1); idx+1 == 0 means
// squarings only.  first_idx is the entry the accumulator is initialised from.
struct Schedule {
    int window = 1;
    int tbl_entries = 1;
    int first_idx = 0;
    std::vector<uint32_t> ops;
    // algorithmic counts, for reporting
    int squarings = 0, multiplies = 0;
};

inline int pick_window(int exp_bits) {
    if (exp_bits <= 8) return 1;
    if (exp_bits <= 24) return 2;
    if (exp_bits <= 80) return 3;
    if (exp_bits <= 240) return 4;
    // measured on the batch-uniform exponents (tools/gpu_ab_window.sh, profiles/r01m_ab_window.txt): w = 6 beats w = 5
    // from ~1000-bit exponents on (2048-bit n: +1.7 % encrypts/s, 3072-bit n: +2.4 %; the 1024 / 1536-bit p - 1 of
    // decrypt: +0.3 % / +1.3 %); w = 7 gains a little more on n but loses on p - 1 and doubles the table again
    if (exp_bits <= 900) return 5;
    return 6;
}

// window of a shared multi-exponentiation ladder (split_core.h:multiexp_split_body): per element bits/chunk
// squarings + bits/w + 2^w - 2 products, and the tables of a limb group are chunk * (2^w - 1) pairs
inline int pick_multi_window(int exp_bits) {
    if (exp_bits <= 4) return 1;
    if (exp_bits <= 24) return 2;
    if (exp_bits <= 110) return 3;
    return 4;
}

inline Schedule build_schedule(const Big& e, int window = 0) {
    const int bits = big_bits(e);
    if (bits == 0) throw std::invalid_argument("exponent must be positive");
    Schedule s;
    s.window = window > 0 ? window : pick_window(bits);
    if (window <= 0 && bits > 240) {  // the long batch-uniform exponents (n, p-1, q-1): PHE_HIP_WINDOW=4..7 overrides w = 5
        if (const char* ev = getenv("PHE_HIP_WINDOW")) {
            const int v = atoi(ev);
            if (v >= 4 && v <= 7) s.window = v;
        }
    }
    s.tbl_entries = 1 << (s.window - 1);
    auto bit = [&](int i) -> uint32_t { return i < 0 ? 0u : (e[(size_t)(i >> 5)] >> (i & 31)) & 1u; };
    int i = bits - 1;
    bool first = true;
    int pending_sq = 0;
    while (i >= 0) {
        if (!bit(i)) {
            ++pending_sq;
            --i;
            continue;
        }
        // longest window [i .. j] of at most `window` bits that ends in a 1
        int j = std::max(i - s.window + 1, 0);
        while (!bit(j)) ++j;
        uint32_t v = 0;
        for (int k = i; k >= j; --k) v = (v << 1) | bit(k);
        const int len = i - j + 1;
        const int idx = (int)(v >> 1);
        if (first) {
            s.first_idx = idx;
            first = false;
        } else {
            int nsq = pending_sq + len;
            while (nsq > 0xffffff) {  // cannot happen for sane sizes; keep the encoding honest
                s.ops.push_back(0xffffffu << 8);
                s.squarings += 0xffffff;
                nsq -= 0xffffff;
            }
            s.ops.push_back(((uint32_t)nsq << 8) | (uint32_t)(idx + 1));
            s.squarings += nsq;
            s.multiplies += 1;
        }
        pending_sq = 0;
        i = j - 1;
    }
    if (pending_sq) {
        s.ops.push_back((uint32_t)pending_sq << 8);
        s.squarings += pending_sq;
    }
    return s;
}

// Constants of the CRT tail (decrypt_tail.h); p < q enforced by the caller like phe/paillier.py:224-229.
struct TailPack {
    int h = 0;
    Big p, q, pinvw, qinvw, hp_r, hq_r, pinvq_r;
    Big hp, hq, pinvq;  // the plain values (for build_tail_wave, which has its own R)
    uint32_t p0inv = 0, q0inv = 0;
};

inline TailPack build_tail(const Big& p_any, const Big& q_any, const Big& hp_any, const Big& hq_any,
                           const Big& pinv_any) {
    TailPack t;
    t.h = (std::max(big_bits(p_any), big_bits(q_any)) + 31) / 32;
    t.p = big_resize(p_any, t.h);
    t.q = big_resize(q_any, t.h);
    if (((t.p[0] & t.q[0]) & 1u) == 0u) throw std::invalid_argument("p and q must be odd");
    if (big_cmp(t.p, t.q) >= 0) throw std::invalid_argument("expected p < q");
    Big hp = big_resize(hp_any, t.h), hq = big_resize(hq_any, t.h), pinv = big_resize(pinv_any, t.h);
    if (big_cmp(hp, t.p) >= 0 || big_cmp(hq, t.q) >= 0 || big_cmp(pinv, t.q) >= 0)
        throw std::invalid_argument("hp/hq/p_inverse out of range");
    t.hp = hp;
    t.hq = hq;
    t.pinvq = pinv;
    t.pinvw = inv_mod_pow2(t.p, t.h);
    t.qinvw = inv_mod_pow2(t.q, t.h);
    t.hp_r = big_shift_mod(hp, 32 * t.h, t.p);
    t.hq_r = big_shift_mod(hq, 32 * t.h, t.q);
    t.pinvq_r = big_shift_mod(pinv, 32 * t.h, t.q);
    t.p0inv = neg_inv32(t.p[0]);
    t.q0inv = neg_inv32(t.q[0]);
    return t;
}

// The same constants for the tail on one WAVEFRONT per ciphertext (split_core.h decrypt_tail_wave_body): rows of H = 64 L
// limbs of 29 bits, R = 2^(29 rows).  L == 0: p, q wider than the whole-wave kernels go (the per-thread tail serves).
struct TailWavePack {
    int L = 0, H = 0, rows = 0;
    std::vector<uint32_t> p, q, pinv, qinv, hp_r, hq_r, pinvq_r;  // H limbs each
    uint32_t p0inv = 0, q0inv = 0;                                 // -p^-1, -q^-1 mod 2^29
    bool ok() const { return L != 0; }
};

inline TailWavePack build_tail_wave(const TailPack& t) {
    TailWavePack W;
    const int bits = big_bits(t.q);  // p < q
    const int need = (bits + 4 + kRadixBits - 1) / kRadixBits;
    for (int L : kS64)
        if ((need + L - 1) / L * L < 64 * L) {
            W.L = L;
            break;
        }
    if (W.L == 0) return W;
    W.H = 64 * W.L;
    W.rows = (need + W.L - 1) / W.L * W.L;
    const int rbits = kRadixBits * W.rows, rw = (rbits + 31) / 32;
    const uint32_t mask = (1u << kRadixBits) - 1u;
    // n^-1 mod R: the inverse modulo 2^(32 rw), cut to rbits bits
    const auto inv_mod_r = [&](const Big& n) {
        Big x = inv_mod_pow2(big_resize(n, rw), rw);
        if (rbits & 31) x[(size_t)rw - 1] &= (1u << (rbits & 31)) - 1u;
        return to_r29(x, W.H);
    };
    W.p = to_r29(t.p, W.H);
    W.q = to_r29(t.q, W.H);
    W.pinv = inv_mod_r(t.p);
    W.qinv = inv_mod_r(t.q);
    W.hp_r = to_r29(big_shift_mod(t.hp, rbits, t.p), W.H);
    W.hq_r = to_r29(big_shift_mod(t.hq, rbits, t.q), W.H);
    W.pinvq_r = to_r29(big_shift_mod(t.pinvq, rbits, t.q), W.H);
    W.p0inv = neg_inv32(t.p[0]) & mask;
    W.q0inv = neg_inv32(t.q[0]) & mask;
    return W;
}

// Public-key side: modulus n^2, exponent n (phe/paillier.py:137, :622), aux = n*R for 1 + n*m.
struct PublicPlan {
    int s1 = 0, s2 = 0;  // ABI widths: limbs of n / of a ciphertext (= 2*s1)
    Big n;               // s1 limbs (32-bit)
    Big nsq32;           // n^2, s2 limbs (32-bit) — host-side scalar work of batched inversion
    ModulusPack nsq;
    SplitPack nsplit;    // pair arithmetic modulo n for encrypt / obfuscate (G == 0: not available)
    // the same for the SCALED modulus n' = k*n, k = -n^-1 mod 2^29 (n' = -1 mod 2^29: quotient digits need no multiply,
    // split_core.h PHE_QUOTIENT_STEP), offered when n' still fits the geometry of n (29 spare bits) and the full-width
    // geometry of n^2 can take the result modulo n'^2 in (G == 0: not offered).  unit_words: 32-bit words of such a row.
    SplitPack nunit;
    int unit_words = 0;
    QuickPack nquick;    // whole-wave plans only: one number on a wave pair
    Schedule exp_n;
};

inline PublicPlan build_public(const uint32_t* n, int n_limbs, int prefer_group = 0) {
    PublicPlan P;
    P.s1 = n_limbs;
    P.s2 = 2 * n_limbs;
    P.n = big_from(n, n_limbs, n_limbs);
    if (big_bits(P.n) < 2) throw std::invalid_argument("n too small");
    Big nsq = big_mul(P.n, P.n);
    P.nsq32 = big_resize(nsq, P.s2);
    P.nsplit = build_split(P.n, 32 * P.s2, prefer_group);
    try {
        P.nsq = build_modulus(nsq, &P.n, 32 * P.s2, prefer_group);
    } catch (const std::invalid_argument&) {
        // n^2 is wider than the widest full-width geometry (keys above ~4170 bits, e.g. the 8192-bit keys of the
        // reference's examples/benchmarks.py:88-90): every job modulo n^2 then runs on the pair form, if that exists
        if (P.nsplit.G == 0) throw;
        P.nsq = ModulusPack();
    }
    P.exp_n = build_schedule(P.n);
    if (P.nsplit.G == 64) P.nquick = build_quick(P.n, 32 * P.s2, quick_lane_width(P.n));
    else if (P.nsplit.G == 16) P.nquick = build_quick(P.n, 32 * P.s2, P.nsplit.L, 16);
    if (P.nsplit.G && P.nsq.G) {
        Big k(1, P.nsplit.n0inv);                       // -n^-1 mod 2^29
        Big nk = big_mul(P.n, k);                       // n' = k*n = -1 (mod 2^29)
        const int bits = big_bits(nk);
        if (bits + 4 <= kRadixBits * P.nsplit.H && 2 * bits <= kRadixBits * P.nsq.S) {
            SplitPack U = build_split(big_resize(nk, (bits + 31) / 32), 32 * P.s2, prefer_group);
            if (U.G == P.nsplit.G && U.L == P.nsplit.L && U.n0inv == 1u) {
                P.nunit = U;
                P.unit_words = ((2 * bits + 31) / 32 + 3) & ~3;  // rows of a multiple of 4 words (16-byte chunks)
            }
        }
    }
    return P;
}

// Private-key side: moduli p^2, q^2 with exponents p-1, q-1 (phe/paillier.py:347, :351) + CRT tail.
struct PrivatePlan {
    int s1 = 0, s2 = 0;
    ModulusPack psq, qsq;  // same L
    SplitPack psplit, qsplit;  // pair arithmetic modulo p / q for the two CRT half-exponentiations
    QuickPack pquick, qquick;  // whole-wave plans only: one number on a wave pair (one lane width for both)
    Schedule exp_p, exp_q;
    TailPack tail;
};

inline PrivatePlan build_private(const uint32_t* p, const uint32_t* q, const uint32_t* hp, const uint32_t* hq,
                                 const uint32_t* p_inverse, int pq_limbs, int n_limbs, int prefer_group = 0) {
    PrivatePlan P;
    P.s1 = n_limbs;
    P.s2 = 2 * n_limbs;
    Big bp = big_from(p, pq_limbs, pq_limbs), bq = big_from(q, pq_limbs, pq_limbs);
    P.tail = build_tail(bp, bq, big_from(hp, pq_limbs, pq_limbs), big_from(hq, pq_limbs, pq_limbs),
                        big_from(p_inverse, pq_limbs, pq_limbs));
    Big psq = big_mul(bp, bp), qsq = big_mul(bq, bq);
    // both halves share one geometry (sized for q^2, the larger); the wide input c (s2 32-bit words) is
    // split at 2^(29 S), so 2*29*S must cover it
    const int min_bits = std::max(big_bits(qsq) + 4, 16 * P.s2);
    P.psq = build_modulus(psq, nullptr, min_bits, prefer_group);
    P.qsq = build_modulus(qsq, nullptr, min_bits, prefer_group);
    if (P.psq.L != P.qsq.L || P.psq.G != P.qsq.G) throw std::invalid_argument("p and q too unbalanced");
    // (prefer_group 0 = automatic: the halves may take one lane per number where p, q are that short)
    P.psplit = build_split(bp, 32 * P.s2, prefer_group == 0 ? 1 : prefer_group);
    P.qsplit = build_split(bq, 32 * P.s2, prefer_group == 0 ? 1 : prefer_group);
    if (P.psplit.G != P.qsplit.G || P.psplit.L != P.qsplit.L) {  // (the halves run as one geometry: the narrowest both have)
        P.psplit = build_split(bp, 32 * P.s2, prefer_group);
        P.qsplit = build_split(bq, 32 * P.s2, prefer_group);
    }
    if (P.psplit.G == 64 && P.qsplit.G == 64) {
        const int Lp = quick_lane_width(bp), Lq = quick_lane_width(bq), Lw = std::max(Lp, Lq);
        if (Lp && Lq) {
            P.pquick = build_quick(bp, 32 * P.s2, Lw);
            P.qquick = build_quick(bq, 32 * P.s2, Lw);
        }
    } else if (P.psplit.G == 16 && P.qsplit.G == 16 && P.psplit.L == P.qsplit.L) {
        P.pquick = build_quick(bp, 32 * P.s2, P.psplit.L, 16);
        P.qquick = build_quick(bq, 32 * P.s2, P.psplit.L, 16);
        if (!(P.pquick.ok() && P.qquick.ok() && P.pquick.scaled.rows == P.qquick.scaled.rows)) P.pquick = P.qquick = QuickPack();
    }
    Big one((size_t)pq_limbs, 0u);
    one[0] = 1;
    Big pm1 = bp, qm1 = bq;
    big_sub_inplace(pm1, one);
    big_sub_inplace(qm1, one);
    P.exp_p = build_schedule(pm1);
    P.exp_q = build_schedule(qm1);
    return P;
}

// ---- a*b mod N as one plain product + one fold against a table (csrc/mul_table.h) ---------------------------------------------
// G = 16 lanes x L limbs, S = 16 L >= (bits(N) + 38) / 29 limbs; P = ceil(bits(N) / 29): the limbs of T = a*b at and above P are
// the D fold digits, row i of the table is W^(P+i) mod N (W = 2^29).  Offered while the table fits the LDS of a CU beside the
// kernel's other areas (keys up to ~2700 bits: L = 5 or 9); mul_tile.h keeps the table in L2 and serves 3072-bit keys (L = 14) too —
// ok() false when neither is offered: mul_io.h's kernels serve.
struct TableMulPack {
    int L = 0, S = 0;
    int split = 0, digits = 0, base = 0, digits_padded = 0, tile_waves = 0;
    double inv = 0.0;                          // W^base / N
    std::vector<uint32_t> n, ncomp, ncomp1;    // S limbs each: N, W^S - N, (W^S - N) * W mod W^S
    std::vector<uint32_t> table;               // digits rows of S words, device layout (mul_table.h table_row_limbs)
    std::vector<uint32_t> table_cols;          // the same rows in the column-block layout [wave][digit][S / 16 words] (mul_tile.h)
    size_t lds_words = 0;
    size_t tile_lds_words = 0;                 // LDS of mul_tile.h's workgroup (no table in it)
    bool ok() const { return L != 0; }                  // some form of the product is offered:
    bool in_lds() const { return lds_words != 0; }      //   mul_table.h (L = 5, 9: the table fits a CU's LDS beside the rest)
    bool tiles() const { return tile_lds_words != 0; }  //   mul_tile.h
};
// lane widths offered: 9 (n^2 of ~1850 ... 2048-bit keys).  5 limbs per lane (1024-bit keys) was built and measured SLOWER than
// the Montgomery kernels there (818 M against 1024 M products/s: 80 limbs for a number of 71, and 5 multiply-adds per table read
// and per shift) — kept compiled for the emulator tests, not offered by the library (build_table_mul's `offer_narrow`)
static const int kTableL[] = {5, 9, 14};
constexpr int kTileWavesHost = 16;  // = mul_tile.h kTileWaves: the waves of its workgroup, the column blocks of table_cols
constexpr size_t kTableLdsLimitBytes = 158 * 1024;  // of the 160 KB of a CU

// waves: the workgroup shape of mul_tile.h the pack is cut for — 16 (S = 16 L columns; also what mul_table.h's limb groups of 16 lanes
// need) or 8 (S = 8 L, tiles only; round 5): n^2 of a 1024-bit key needs 72 columns = 8 x 9 exactly, where 16 x 5 pads to 80
constexpr int kTileL8 = 9;  // the lane width compiled for 8-wave workgroups
inline TableMulPack build_table_mul(const Big& N_any, int word_limbs, bool offer_narrow = true, int waves = kTileWavesHost) {
    TableMulPack T;
    const int bits = big_bits(N_any);
    if (bits < 64 || bits > 32 * word_limbs || (waves != 16 && waves != 8)) return T;
    const int need = (bits + 38 + kRadixBits - 1) / kRadixBits;
    int L = 0;
    if (waves == 8) {
        if (8 * kTileL8 >= need) L = kTileL8;
    } else {
        for (int cand : kTableL)
            if (16 * cand >= need) {
                L = cand;
                break;
            }
    }
    if (!L || (!offer_narrow && (L < 9 || waves * L - need >= 16))) return T;  // (the library: only where the lanes are well filled)
    const int S = waves * L, P = (bits + kRadixBits - 1) / kRadixBits, n_lo = S - P;
    // limbs the high half of a product of two `word_limbs`-word numbers can have, on top of the S low ones
    const int hi_limbs = std::max(0, (2 * 32 * word_limbs + kRadixBits - 1) / kRadixBits - S);
    const int D = n_lo + hi_limbs;
    // (16 waves: the quotient estimate reads limbs P - 2 ... P + 1 and the settle rows start at S - 2: P <= S - 2; 8 waves: P = S - 1 is
    //  taken too — the estimate moves down a limb, the settle rows start at S: mul_tile.h TileShape)
    if (n_lo < (waves == 8 ? 1 : 2) || n_lo > 16 || hi_limbs > S) return T;
    // LDS of the 512-thread workgroup (mul_table.h table_lds_words): table | 3 constant rows | 32 digit rows | 8 x 2 staging areas
    const int k_words = (S * kRadixBits + 31) / 32, k_vec = (k_words + 1 + 63) / 64;
    size_t lds_words = (size_t)D * S + 3 * (size_t)S + 32 * (size_t)(S + 16) + 8 * 2 * (size_t)(256 * k_vec);
    if (lds_words * 4 > kTableLdsLimitBytes || L > 9 || waves != 16) lds_words = 0;  // (mul_table.h is compiled for 16 lanes x L = 5, 9)
    // mul_tile.h TileShape<L, W>::kLdsWords: tile buffer (the settle's digit rows inside it) | product carries | top columns | fold carries | 3 constant rows
    const size_t settle_row0 = waves == 16 ? (size_t)(S - 2) : (size_t)S;
    const size_t tile_rows = std::max<size_t>(2 * (size_t)S + 1, settle_row0 + (size_t)(S + 4));
    size_t tile_lds_words = tile_rows * 64 + 2 * 2 * (size_t)waves * 64 + 64 * 16 + 2 * (size_t)waves * 64 + 3 * (size_t)S;
    if (tile_lds_words * 4 > kTableLdsLimitBytes) tile_lds_words = 0;
    if (!lds_words && !tile_lds_words) return T;
    const int w32 = (kRadixBits * S + 31) / 32 + 1;  // words that hold W^S
    const Big N = big_resize(N_any, w32);
    if ((N[0] & 1u) == 0u) return T;
    T.n = to_r29(N, S);
    {
        Big top((size_t)w32, 0u);  // W^S
        top[(size_t)((kRadixBits * S) >> 5)] = 1u << ((kRadixBits * S) & 31);
        Big nc = top;
        big_sub_inplace(nc, N);
        T.ncomp = to_r29(nc, S);
        T.ncomp1.assign((size_t)S, 0u);
        for (int k = 1; k < S; ++k) T.ncomp1[(size_t)k] = T.ncomp[(size_t)k - 1];  // * W, the limb shifted out of W^S dropped
    }
    // rows W^(P+i) mod N
    Big one((size_t)w32, 0u);
    one[0] = 1;
    Big c = big_shift_mod(one, kRadixBits * P, N);
    constexpr int kFullMax = 16;
    const int full = L / 4, rem = L % 4;
    (void)kFullMax;
    T.table.assign((size_t)D * S, 0u);
    const int cw = S / waves;  // columns of one wave of mul_tile.h's fold
    T.digits_padded = (D + 7) & ~7;
    const int d_rows = T.digits_padded + 4;  // (mul_tile.h kFoldPadRows: zero rows for the fold's look-ahead)
    T.table_cols.assign((size_t)waves * d_rows * cw, 0u);
    T.tile_waves = waves;
    for (int i = 0; i < D; ++i) {
        const std::vector<uint32_t> limbs = to_r29(c, S);
        for (int w = 0; w < waves; ++w)
            for (int k = 0; k < cw; ++k) T.table_cols[((size_t)w * d_rows + (size_t)i) * cw + k] = limbs[(size_t)(w * cw + k)];
        uint32_t* row = T.table.data() + (size_t)i * S;
        for (int g = 0; g < 16 && waves == 16; ++g) {  // (mul_table.h's layout: limb groups of 16 lanes)
            for (int q = 0; q < full; ++q)
                for (int e = 0; e < 4; ++e) row[q * 64 + 4 * g + e] = limbs[(size_t)(g * L + 4 * q + e)];
            for (int r = 0; r < rem; ++r) row[full * 64 + g * rem + r] = limbs[(size_t)(g * L + 4 * full + r)];
        }
        c = big_shift_mod(c, kRadixBits, N);
    }
    // W^base / N as a double: N = top64 * 2^(bits - 64) up to 2^-63 relative
    {
        uint64_t top64 = 0;
        for (int b = 0; b < 64; ++b) {
            const int bit = bits - 1 - b;
            top64 = (top64 << 1) | ((N[(size_t)(bit >> 5)] >> (bit & 31)) & 1u);
        }
        T.base = std::min(P - 2, S - 4);  // the four limbs the estimate reads end at the top limb a y < 2^37 N can have
        T.inv = __builtin_ldexp(1.0 / (double)top64, kRadixBits * T.base - (bits - 64));
    }
    T.L = L;
    T.S = S;
    T.split = P;
    T.digits = D;
    T.lds_words = lds_words;
    T.tile_lds_words = tile_lds_words;
    return T;
}

// scalar inverse of a residue modulo an odd N (binary extended Euclid); false if gcd != 1
inline bool big_invert_odd(const Big& a_in, const Big& N, Big& out) {
    const size_t w = N.size();
    Big u = a_in, v = N, x1(w, 0u), x2(w, 0u);
    x1[0] = 1;
    auto is_one = [](const Big& x) {
        if (x[0] != 1) return false;
        for (size_t i = 1; i < x.size(); ++i)
            if (x[i]) return false;
        return true;
    };
    auto halve_mod = [&](Big& x) {  // x <- x/2 mod N
        uint32_t carry = 0;
        if (x[0] & 1u) carry = big_add_inplace(x, N);
        for (size_t i = 0; i + 1 < w; ++i) x[i] = (x[i] >> 1) | (x[i + 1] << 31);
        x[w - 1] = (x[w - 1] >> 1) | (carry << 31);
    };
    auto shr1 = [&](Big& x) {
        for (size_t i = 0; i + 1 < w; ++i) x[i] = (x[i] >> 1) | (x[i + 1] << 31);
        x[w - 1] >>= 1;
    };
    if (big_is_zero(u)) return false;
    while (!is_one(u) && !is_one(v)) {
        while ((u[0] & 1u) == 0u) { shr1(u); halve_mod(x1); }
        while ((v[0] & 1u) == 0u) { shr1(v); halve_mod(x2); }
        if (big_cmp(u, v) >= 0) {
            big_sub_inplace(u, v);
            if (big_sub_inplace(x1, x2)) big_add_inplace(x1, N);
            if (big_is_zero(u)) return false;  // gcd = v != 1
        } else {
            big_sub_inplace(v, u);
            if (big_sub_inplace(x2, x1)) big_add_inplace(x2, N);
        }
    }
    out = is_one(u) ? x1 : x2;
    return true;
}


// which (G, L) the split translation units instantiate (kernels_s*.hip): the CRT lift runs there on the full-width
// geometry of q^2
inline bool split_part_holds(int G, int L) {
    const int* list = G == 16 ? kS16 : G == 8 ? kS8 : G == 4 ? kS4 : G == 2 ? kS2 : nullptr;  // (no lift on one lane: q^2 never fits)
    const auto len = [](const auto& a) { return (int)(sizeof(a) / sizeof(a[0])); };
    const int count = G == 16 ? len(kS16) : G == 8 ? len(kS8) : G == 4 ? len(kS4) : G == 2 ? len(kS2) : 0;  // (the lift has no whole-wave form: G = 64 is not asked here)
    for (int i = 0; i < count; ++i)
        if (list[i] == L) return true;
    return false;
}

// Encryption by the key owner (split_core.h:crt_lift_body): K = (p^2)^-1 mod q^2 as K*R and (q^2 - K)*R mod q^2 (R =
// 2^(29 S), S the limbs of q^2's full-width geometry), and p^2, as rows of S limbs.  false: no inverse (p^2, q^2 not coprime)
struct OwnerLift {
    std::vector<uint32_t> kr, nkr, psq;
};
inline bool build_owner_lift(const Big& bp, const Big& bq, int S, OwnerLift& W) {
    const Big qsq = big_mul(bq, bq), psq = big_mul(bp, bp);
    const int w32 = (big_bits(qsq) + 31) / 32;
    const Big N = big_resize(qsq, w32);
    const Big P2 = big_resize(psq, w32);  // p < q: p^2 < q^2
    Big K;
    if (!big_invert_odd(P2, N, K)) return false;
    Big NK = N;
    big_sub_inplace(NK, K);
    const int rbits = kRadixBits * S;
    W.kr = to_r29(big_shift_mod(K, rbits, N), S);
    W.nkr = to_r29(big_shift_mod(NK, rbits, N), S);
    W.psq = to_r29(P2, S);
    return true;
}

}  // namespace host
}  // namespace phe
