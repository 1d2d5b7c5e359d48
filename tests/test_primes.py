"""Batched primality / prime search (csrc/primality.h, kernels_mr.hip, phe/primes.py).

What the reference does: getprimeover (phe/util.py:106-124) = random N-bit start with the top bit set, then
gmpy2.next_prime(start) — libgmp's mpz_nextprime — or, without gmpy2, is_prime -> miller_rabin (phe/util.py:381-443).
Checks: the per-row-modulus Miller-Rabin launch against a Python restatement of phe/util.py:411-443's witness loop
(primes, Carmichael numbers, strong pseudoprimes to base 2, mixed bit lengths in one wavefront); next_primes against
libgmp's mpz_nextprime on seeded starts; key pairs against generate_paillier_keypair's acceptance rules
(phe/paillier.py:37-68).  CPU run: the device header on the wave emulator; -m gpu: the kernels through the C-ABI."""
import random
import sys

import numpy as np
import pytest

from conftest import PKG

if PKG not in sys.path:
    sys.path.insert(0, PKG)

from oracle.paillier_oracle import ints_to_limbs  # noqa: E402


def strong_probable_prime(n, a):
    """one witness round of phe/util.py:411-443 (miller_rabin), base given"""
    d, s = n - 1, 0
    while d % 2 == 0:
        d, s = d // 2, s + 1
    x = pow(a, d, n)
    if x == 1 or x == n - 1:
        return True
    for _ in range(s - 1):
        x = x * x % n
        if x == n - 1:
            return True
    return False


KNOWN = [(2047, 2), (2047, 3), (3277, 2), (3277, 5), (561, 2), (1105, 2), (1729, 7), (65537, 3), (2 ** 31 - 1, 7),
         (2 ** 61 - 1, 2), (2 ** 89 - 1, 3), (2 ** 127 - 1, 3), (2 ** 16 + 1, 2), (1 + 7 * 2 ** 20, 3), (5, 2), (5, 3), (9, 2)]


def _cases(rng, limbs, bitsets, per):
    ns, bs = [], []
    for bits in bitsets:
        for _ in range(per):
            n = rng.getrandbits(bits) | 1 | (1 << (bits - 1))
            ns.append(max(n, 5))
            bs.append(rng.randrange(2, ns[-1] - 1))
    for n, a in KNOWN:
        if n < 2 ** (32 * limbs):
            ns.append(n)
            bs.append(a)
    return ns, bs


@pytest.fixture(scope="module")
def emu():
    from emu_lib import Emu
    return Emu()


@pytest.mark.parametrize("limbs,bitsets", [(1, (20, 31, 32)), (2, (40, 64)), (4, (100, 128)), (8, (200, 256))])
def test_miller_rabin_rows_on_the_emulator(emu, limbs, bitsets):
    rng = random.Random(limbs)
    ns, bs = _cases(rng, limbs, bitsets, 10)
    got = emu.miller_rabin(ints_to_limbs(ns, limbs), ints_to_limbs(bs, limbs))
    assert got.tolist() == [strong_probable_prime(n, a) for n, a in zip(ns, bs)]


def test_prime_search_matches_gmp_nextprime_on_the_emulator(monkeypatch, c_oracle):
    import emu_backend
    emu_backend.install(monkeypatch)
    from phe import primes
    rng = random.Random(9)
    starts = [rng.getrandbits(bits) | (1 << (bits - 1)) for bits in (64, 96, 128)]
    got = primes.next_primes(starts, rounds=4)
    assert got == [c_oracle.next_prime(s) for s in starts]        # gmpy2.next_prime (phe/util.py:116)
    assert primes.is_probable_prime(2 ** 127 - 1, 6) and not primes.is_probable_prime(2 ** 127 + 1, 6)
    assert not primes.is_probable_prime(561, 4) and primes.is_probable_prime(7919, 3)
    (pub, priv), = primes.generate_paillier_keypairs(1, 128)
    assert pub.n.bit_length() == 128 and priv.p != priv.q and priv.p * priv.q == pub.n      # phe/paillier.py:57-66
    with pytest.raises(ValueError):
        primes.next_primes([1000])


# ---- GPU -------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("limbs,bitsets", [(1, (20, 32)), (4, (100, 128)), (16, (500, 512)), (32, (1000, 1024)),
                                           (48, (1536,)), (64, (2048,))])
def test_miller_rabin_kernel(limbs, bitsets, c_oracle):
    from phe import _native
    rng = random.Random(limbs)
    ns, bs = _cases(rng, limbs, bitsets, 40 if limbs <= 32 else 12)
    if limbs >= 16:                                               # real primes of that size, found by libgmp
        for bits in bitsets:
            p = c_oracle.next_prime(rng.getrandbits(bits - 1) | (1 << (bits - 2)))
            ns += [p, p]
            bs += [2, rng.randrange(2, p - 1)]
    got = _native.miller_rabin(ints_to_limbs(ns, limbs), ints_to_limbs(bs, limbs))
    assert got.tolist() == [strong_probable_prime(n, a) for n, a in zip(ns, bs)]
    with pytest.raises(ValueError):
        _native.miller_rabin(ints_to_limbs([15, 16], limbs), ints_to_limbs([2, 2], limbs))     # even candidate
    with pytest.raises(ValueError):
        _native.miller_rabin(ints_to_limbs([15], limbs), ints_to_limbs([14], limbs))           # base > n - 2


@pytest.mark.gpu
def test_prime_search_and_keypairs_on_gpu(c_oracle):
    from phe import primes
    rng = random.Random(11)
    starts = [rng.getrandbits(bits) | (1 << (bits - 1)) for bits in (256, 512, 1024, 1024, 1536)]
    assert primes.next_primes(starts) == [c_oracle.next_prime(s) for s in starts]
    pairs = primes.generate_paillier_keypairs(3, 1024)
    for pub, priv in pairs:
        assert pub.n.bit_length() == 1024 and priv.p != priv.q and priv.p * priv.q == pub.n
        assert c_oracle.is_probable_prime(priv.p) and c_oracle.is_probable_prime(priv.q)
        vals = [0.5, -3.25, 12345.0]
        assert priv.decrypt_batch(pub.encrypt_batch(np.array(vals))) == vals


@pytest.mark.gpu
def test_generate_paillier_keypair_uses_the_batched_search(c_oracle):
    """the reference's entry point (phe/paillier.py:37-68) on a GPU box: same contract, primes from phe.primes"""
    import time
    from phe import keys, paillier
    assert keys._gpu_prime_search_available()
    t0 = time.perf_counter()
    pub, priv = paillier.generate_paillier_keypair(n_length=2048)
    elapsed = time.perf_counter() - t0
    assert pub.n.bit_length() == 2048 and priv.p * priv.q == pub.n and priv.p != priv.q
    assert priv.p.bit_length() == priv.q.bit_length() == 1024
    assert c_oracle.is_probable_prime(priv.p) and c_oracle.is_probable_prime(priv.q)
    assert priv.decrypt(pub.encrypt(-12.5)) == -12.5
    ring = paillier.PaillierPrivateKeyring()
    pub2, priv2 = paillier.generate_paillier_keypair(ring, 1024)
    assert ring[pub2] == priv2 and pub2.n.bit_length() == 1024
    assert elapsed < 5.0
