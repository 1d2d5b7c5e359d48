"""Prime search on the GPU: the batched form of getprimeover (phe/util.py:106-124 of the reference).

The reference draws a random N-bit start with the top bit set and takes gmpy2.next_prime(start) (or, without gmpy2,
walks the odd numbers with is_prime -> miller_rabin, phe/util.py:381-443): one strong-probable-prime test after the
other.  Here a whole window of candidates above the start is sieved by small primes on the host (numpy) and every
survivor is tested to base 2 in ONE launch of the per-row-modulus Miller-Rabin kernel (csrc/primality.h); the first
survivor that passes is confirmed with `rounds` random bases in a second launch (the same candidate in every row).
The result is the smallest probable prime above the start — what gmpy2.next_prime returns (tests compare it with
libgmp's mpz_nextprime on seeded starts).  Several searches can share the launches (`next_primes`): that is where a
GPU pays — a single 1024-bit search is latency-bound and takes about as long as gmpy2 on one core, a batch of keys
costs little more than one.

`generate_paillier_keypairs(count, n_length)` is generate_paillier_keypair (phe/paillier.py:37-68) for many keys: the
same acceptance rules (p != q, n = p*q has exactly n_length bits), primes from `getprimeover_batch`.
"""
import random
import secrets

import numpy as np

from . import _native

_SMALL_PRIMES = None


def _small_primes(limit=1 << 13):
    global _SMALL_PRIMES
    if _SMALL_PRIMES is None:
        sieve = np.ones(limit, dtype=bool)
        sieve[:2] = False
        for i in range(2, int(limit ** 0.5) + 1):
            if sieve[i]:
                sieve[i * i::i] = False
        _SMALL_PRIMES = np.nonzero(sieve)[0].astype(np.int64)
    return _SMALL_PRIMES


def _survivors(start, span):
    """offsets d in [1, span] with start + d odd and free of prime factors below 2^13 (start >= 2^17)"""
    primes = _small_primes()[1:]                                  # odd primes
    first = 1 if start % 2 == 0 else 2                            # smallest offset that makes start + d odd
    offs = np.arange(first, span + 1, 2, dtype=np.int64)
    alive = np.ones(len(offs), dtype=bool)
    rem = np.array([start % int(p) for p in primes.tolist()], dtype=np.int64)
    # start + d = 0 (mod p)  <=>  d = -start (mod p); offsets are first, first + 2, ...: index i has d = first + 2 i
    for p, r in zip(primes.tolist(), rem.tolist()):
        d0 = (-r) % p                                            # smallest d >= 0
        # solve first + 2 i = d0 (mod p)  ->  i = (d0 - first) * inv(2) (mod p)
        i0 = ((d0 - first) * ((p + 1) // 2)) % p
        alive[i0::p] = False
    return offs[alive]


def _rows(values, limbs):
    return _native.ints_to_limbs(values, limbs)


def miller_rabin_batch(candidates, bases, device=0):
    """strong-probable-prime test of candidates[i] to bases[i] (ints: odd > 3, 2 <= base <= n - 2) -> list of bool"""
    if not candidates:
        return []
    limbs = max(1, (max(c.bit_length() for c in candidates) + 31) // 32)
    return _native.miller_rabin(_rows(candidates, limbs), _rows(bases, limbs), device).tolist()


def is_probable_prime(n, rounds=25, device=0, rng=None):
    """Miller-Rabin with `rounds` random bases in one launch (plus the small-prime shortcuts of phe/util.py:395-408)."""
    if n < 2:
        return False
    for p in _small_primes()[:64].tolist():
        if n == p:
            return True
        if n % p == 0:
            return False
    rng = rng or random.SystemRandom()
    bases = [rng.randrange(2, n - 1) for _ in range(rounds)]
    return all(miller_rabin_batch([n] * rounds, bases, device))


def next_primes(starts, rounds=25, device=0, span=None, rng=None):
    """[smallest probable prime > s for s in starts] — all searches share the launches.  starts >= 2^17."""
    starts = [int(s) for s in starts]
    if any(s < (1 << 17) for s in starts):
        raise ValueError("starts below 2^17 are not worth a launch: use phe.util.is_prime")
    rng = rng or random.SystemRandom()
    result = [None] * len(starts)
    base_off = [0] * len(starts)                                  # offsets already searched
    while any(r is None for r in result):
        todo = [i for i, r in enumerate(result) if r is None]
        cands, owner = [], []
        for i in todo:
            s = starts[i] + base_off[i]
            width = span or max(1024, 8 * s.bit_length())        # ~11 x the expected gap ln(s); a miss just widens the search
            offs = _survivors(s, width)
            cands += [s + int(d) for d in offs.tolist()]
            owner += [i] * len(offs)
            base_off[i] += width
        hits = miller_rabin_batch(cands, [2] * len(cands), device) if cands else []
        # first base-2 survivor per search, confirmed with random bases (all confirmations in one launch)
        firsts = {}
        for c, i, h in zip(cands, owner, hits):
            if h and i not in firsts:
                firsts[i] = c
        while firsts:
            order = sorted(firsts)
            ns = [firsts[i] for i in order for _ in range(rounds)]
            bs = [rng.randrange(2, n - 1) for n in ns]
            ok = miller_rabin_batch(ns, bs, device)
            nxt = {}
            for k, i in enumerate(order):
                if all(ok[k * rounds:(k + 1) * rounds]):
                    result[i] = firsts[i]
                else:                                            # a base-2 pseudoprime: move to the next base-2 survivor
                    later = [c for c, o, h in zip(cands, owner, hits) if o == i and h and c > firsts[i]]
                    if later:
                        nxt[i] = later[0]
            firsts = nxt
    return result


def next_prime(start, rounds=25, device=0):
    return next_primes([start], rounds, device)[0]


def getprimeover_batch(N, count, device=0):
    """`count` random N-bit probable primes: secrets.randbits(N) with the top bit set, then the next prime — the gmpy2
    branch of getprimeover (phe/util.py:113-116), `count` searches at once."""
    if N < 18:
        raise ValueError("N too small for the batched search")
    out = []
    while len(out) < count:
        starts = [secrets.randbits(N) | (1 << (N - 1)) for _ in range(count - len(out))]
        out += [p for p in next_primes(starts, device=device) if p.bit_length() == N]
    return out


def generate_paillier_keypairs(count, n_length=None, device=0):
    """`count` key pairs, each exactly as generate_paillier_keypair makes one (phe/paillier.py:37-68): p, q of
    n_length // 2 bits, p != q, bit_length(p * q) == n_length."""
    from .keys import DEFAULT_KEYSIZE, PaillierPrivateKey, PaillierPublicKey
    n_length = n_length or DEFAULT_KEYSIZE
    pairs = []
    while len(pairs) < count:
        need = count - len(pairs)
        primes = getprimeover_batch(n_length // 2, 2 * need, device)
        for p, q in zip(primes[0::2], primes[1::2]):
            if p != q and (p * q).bit_length() == n_length:
                pub = PaillierPublicKey(p * q)
                pairs.append((pub, PaillierPrivateKey(pub, p, q)))
    return pairs[:count]
