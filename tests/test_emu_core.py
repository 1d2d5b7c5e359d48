"""CPU tests of the code the GPU runs: csrc/mont_core.h, csrc/decrypt_tail.h and csrc/key_setup.h are
compiled for the host on a fiber-based wavefront emulator (tests/emu/) and compared, limb for limb,
with the golden fixtures (real reference) and the libgmp oracle."""
import ctypes
import json
import os
import random

import numpy as np
import pytest

from conftest import load_golden, load_kat
from emu_lib import Emu, from_r29, to_r29
from oracle.paillier_oracle import int_to_limbs, ints_to_limbs, limbs_to_int, limbs_to_ints


def H(x):
    return int(x, 16)


@pytest.fixture(scope="module")
def emu():
    return Emu()


GEOMETRIES = [(16, 1), (16, 2), (16, 3), (16, 5), (16, 7), (16, 9), (16, 14), (16, 18), (8, 5), (8, 9), (8, 14), (8, 18),
              (4, 9), (4, 18), (4, 36), (8, 27), (4, 27), (2, 18), (2, 36)]


@pytest.mark.parametrize("G,L", GEOMETRIES)
def test_montmul_groups(emu, G, L):
    """One radix-2^29 Montgomery product per limb group: result == a*b/R (mod N), < 2N, limbs < 2^29 + 2^8."""
    rng = random.Random(100 * G + L)
    S = G * L
    R = 1 << (29 * S)
    per = 64 // G
    top = 29 * S - 4                                            # largest modulus this geometry accepts
    moduli = [rng.getrandbits(top) | 1 | (1 << (top - 1)),      # widest
              126869 ** 2,                                      # tiny modulus in a wide container
              (1 << (top - 1)) + 1]
    if L <= 31:
        moduli.append((1 << top) - 1)     # all-ones limbs: every carry path (L > 31 is never chosen for such a modulus)
    for N in moduli:
        n0inv = (-pow(N, -1, 1 << 29)) % (1 << 29)
        a = [rng.randrange(0, R) for _ in range(per)]           # multiplier: any value below R
        b = [rng.randrange(0, N) for _ in range(per)]
        a[0], b[0] = R - 1, N - 1
        a[1], b[per - 1] = 0, 0
        aa = np.stack([to_r29(x, S) for x in a])
        bb = np.stack([to_r29(x, S) for x in b])
        out = emu.montmul(G, L, aa, bb, to_r29(N, S), n0inv)
        rinv = pow(R, -1, N)
        for x, y, limbs in zip(a, b, out):
            v = from_r29(limbs)
            assert v % N == x * y * rinv % N
            assert v < 2 * N and int(limbs.max()) < (1 << 29) + (1 << 8)
        # both operands in [N, 2N) and almost-normalised digits: the lazy-reduction invariant is closed
        a2 = [N + rng.randrange(0, N) for _ in range(per)]
        b2 = [N + rng.randrange(0, N) for _ in range(per)]
        bb2 = np.stack([to_r29(x, S) for x in b2])
        aa2 = np.stack([to_r29(x, S) for x in a2])
        out = emu.montmul(G, L, aa2, bb2, to_r29(N, S), n0inv)
        for x, y, limbs in zip(a2, b2, out):
            v = from_r29(limbs)
            assert v % N == x * y * rinv % N and v < 2 * N


def test_dense_modulus_falls_back_to_unconditionally_safe_geometry(emu):
    """L = 36 keeps the 64-bit column sums below 2^64 only if no lane of the modulus is unusually dense;
    key_setup.h checks that per modulus and otherwise stays at L <= 31 — and the result is still exact."""
    emu.set_group(0)
    rng = random.Random(36)
    dense = (1 << 4090) - 1 - (rng.getrandbits(64) << 1)         # ~every 29-bit limb at its maximum
    typical = rng.getrandbits(4090) | 1 | (1 << 4089)
    assert emu.modulus_geometry(int_to_limbs(typical, 128)) == (4, 36)
    G, L = emu.modulus_geometry(int_to_limbs(dense, 128))
    assert L <= 31 and G * L >= 142
    a = [dense - 1, dense - 2, rng.randrange(dense), 1]
    b = [dense - 1, dense - 3, rng.randrange(dense), dense - 1]
    got = emu.mulmod(int_to_limbs(dense, 128), ints_to_limbs(a, 128), ints_to_limbs(b, 128))
    assert limbs_to_ints(got) == [x * y % dense for x, y in zip(a, b)]
    # the same worst-case operands under a typical modulus, on the 4x36 split
    a = [typical - 1, typical - 2, (1 << 4089) - 1, rng.randrange(typical)]
    got = emu.mulmod(int_to_limbs(typical, 128), ints_to_limbs(a, 128), ints_to_limbs(a, 128))
    assert limbs_to_ints(got) == [x * x % typical for x in a]


@pytest.mark.parametrize("key_bits", [256, 1024])
def test_public_constants(emu, key_bits):
    g = load_golden(key_bits)
    n = H(g["n"])
    s1 = key_bits // 32
    for group in (0, 2, 4, 8, 16):
        emu.set_group(group)
        k = emu.public_constants(int_to_limbs(n, s1))
        S = k["S"]
        N = n * n
        R = 1 << (29 * S)
        assert S == k["G"] * k["L"] and 29 * S >= N.bit_length() + 4 and 29 * S >= 64 * s1
        assert from_r29(k["n"]) == N
        assert from_r29(k["r1"]) == R % N
        assert from_r29(k["r2"]) == R * R % N
        assert from_r29(k["r3"]) == R * R * R % N
        assert from_r29(k["aux"]) == n * R % N
        assert (k["n0inv"] * N + 1) % (1 << 29) == 0
        # schedule cost vs the canonical E(t) = t + t/6 + 16 of SURVEY.md 8(d)
        assert k["squarings"] <= key_bits and k["multiplies"] <= key_bits // 5 + 2
    emu.set_group(0)


def test_reference_kat(emu):
    k = load_kat()
    n = int_to_limbs(k["n"], 1)
    c = emu.encrypt(n, ints_to_limbs([k["m"], 1], 1), ints_to_limbs([k["r"], 1], 1))
    assert limbs_to_ints(c) == [k["c"], k["encrypt_1_r_1"]]
    one = lambda x: int_to_limbs(x, 1)
    m = emu.decrypt(one(k["p"]), one(k["q"]), one(k["hp"]), one(k["hq"]), one(k["p_inverse"]), 1, c)
    assert limbs_to_ints(m) == [k["m"], 1]


# engine "split": the split-modulus kernels (csrc/split_core.h, the product default); "full": mont_core.h on n^2, p^2, q^2
@pytest.mark.parametrize("engine,key_bits,count,group", [
    ("split", 256, None, 16), ("split", 256, None, 8), ("split", 1024, 5, 0), ("split", 1024, 3, 16),
    ("split", 1024, 3, 4), ("split", 1024, 3, 8), ("split", 2048, 3, 0), ("split", 2048, 2, 8), ("split", 2048, 2, 16),
    ("split", 3072, 2, 0), ("split", 3072, 1, 8),
    # the whole wavefront as one limb group (G = 64, the latency rung): numbers of 36 / 72 / 108 limbs on 64 x 1 / 2 / 2 lanes
    ("split", 256, None, 64), ("split", 1024, 3, 64), ("split", 2048, 2, 64), ("split", 3072, 1, 64),
    ("full", 256, None, 16), ("full", 256, None, 8), ("full", 1024, 3, 0), ("full", 1024, 2, 4), ("full", 2048, 2, 0),
    ("full", 3072, 1, 0)])
def test_golden_through_emulator(emu, engine, key_bits, count, group):
    emu.set_engine(engine == "split")
    emu.set_group(group)
    g = load_golden(key_bits)
    s1, s2, h = key_bits // 32, key_bits // 16, key_bits // 64
    n = int_to_limbs(H(g["n"]), s1)
    enc = g["raw_encrypt"]
    if count:  # edge cases first (m = 0, 1, ..., n-1, n, n+1), then a few random
        enc = enc[7:7 + count]
    c = emu.encrypt(n, ints_to_limbs([H(e["m"]) for e in enc], s1), ints_to_limbs([H(e["r"]) for e in enc], s1))
    assert limbs_to_ints(c) == [H(e["c"]) for e in enc]

    dec = g["raw_decrypt"]
    if count:
        dec = dec[-count:]
    key = [int_to_limbs(H(g[k]), h) for k in ("p", "q", "hp", "hq", "p_inverse")]
    m = emu.decrypt(*key, s1, ints_to_limbs([H(e["c"]) for e in dec], s2))
    assert limbs_to_ints(m) == [H(e["m"]) for e in dec]
    emu.set_engine(True)
    emu.set_group(0)


@pytest.mark.parametrize("engine", ["split", "full"])
def test_homomorphic_ops_through_emulator(emu, engine):
    emu.set_engine(engine == "split")
    g = load_golden(256)
    s1, s2 = 8, 16
    n_int = H(g["n"])
    N = int_to_limbs(n_int * n_int, s2)
    add = g["raw_add"]
    got = emu.mulmod(N, ints_to_limbs([H(e["a"]) for e in add], s2), ints_to_limbs([H(e["b"]) for e in add], s2))
    assert limbs_to_ints(got) == [H(e["out"]) for e in add]

    obf = g["obfuscate"]
    got = emu.obfuscate(int_to_limbs(n_int, s1), ints_to_limbs([H(e["c_in"]) for e in obf], s2),
                        ints_to_limbs([H(e["r"]) for e in obf], s1))
    assert limbs_to_ints(got) == [H(e["c_out"]) for e in obf]

    # positive-branch scalar multiplications are plain powmods (phe/paillier.py:751)
    max_int = H(g["max_int"])
    mul = [e for e in g["raw_mul"] if H(e["s"]) < n_int - max_int]
    n_arr = int_to_limbs(n_int, s1)
    got = emu.powmod_n2(n_arr, ints_to_limbs([H(e["c"]) for e in mul], s2), ints_to_limbs([H(e["s"]) for e in mul], s1))
    assert limbs_to_ints(got) == [H(e["out"]) for e in mul]
    # negative branch: powmod(invert(c), n - s) (phe/paillier.py:745-749); the inverse comes from Python here
    neg = [e for e in g["raw_mul"] if H(e["s"]) >= n_int - max_int]
    bases = [pow(H(e["c"]), -1, n_int * n_int) for e in neg]
    got = emu.powmod_n2(n_arr, ints_to_limbs(bases, s2), ints_to_limbs([n_int - H(e["s"]) for e in neg], s1))
    assert limbs_to_ints(got) == [H(e["out"]) for e in neg]
    emu.set_engine(True)


@pytest.mark.parametrize("key_bits,group", [(1024, 0), (1024, 8), (2048, 0), (1024, 64), (2048, 64)])
def test_powmod_n2_split_engine(emu, key_bits, group):
    """per-element exponents through k_modexp_var_split's body: 0, 1, short, long and full-width exponents; bases
    0, 1, n^2 - 1, multiples of n and random residues"""
    emu.set_engine(True)
    emu.set_group(group)
    g = load_golden(key_bits)
    n_int = H(g["n"])
    nsq = n_int * n_int
    s1, s2 = key_bits // 32, key_bits // 16
    rng = random.Random(key_bits + group)
    bases = [0, 1, nsq - 1, n_int, n_int * (n_int - 1), rng.randrange(nsq), rng.randrange(nsq), rng.randrange(nsq)]
    exps = [5, 0, 3, 2, 1, rng.getrandbits(56), 16 ** 9, rng.getrandbits(32 * s1)]
    if key_bits > 1024:
        bases, exps = bases[:6], exps[:6]
    got = emu.powmod_n2(int_to_limbs(n_int, s1), ints_to_limbs(bases, s2), ints_to_limbs(exps, s1))
    assert limbs_to_ints(got) == [pow(b, e, nsq) for b, e in zip(bases, exps)]
    emu.set_group(0)


def test_powmod_var_mixed_lengths(emu):
    rng = random.Random(7)
    S = 16
    N_int = (rng.getrandbits(500) | 1 | (1 << 499))
    N = int_to_limbs(N_int, S)
    bases = [rng.randrange(1, N_int) for _ in range(7)]
    exps = [0, 1, 2, 16 ** 7, rng.getrandbits(56), rng.getrandbits(200), (1 << 64) - 1]
    got = emu.powmod_var(N, ints_to_limbs(bases, S), ints_to_limbs(exps, 8))
    assert limbs_to_ints(got) == [pow(b, e, N_int) for b, e in zip(bases, exps)]


@pytest.mark.parametrize("key_bits,group", [(256, 0), (1024, 8), (2048, 8), (2048, 16), (3072, 16)])
def test_staged_products_equal_the_plain_body(emu, key_bits, group):
    """csrc/mul_io.h (rows copied global -> LDS as 16-byte chunks, limbs re-sliced from LDS, one- and two-product forms,
    row-constant operand) against mont_core.h:mulmod_body and Python integers"""
    import ctypes
    from emu_lib import P
    g = load_golden(key_bits)
    n = H(g["n"])
    N, s2 = n * n, key_bits // 16
    rng = random.Random(key_bits)
    B = 11
    a = [rng.randrange(N) for _ in range(B)]
    b = [rng.randrange(N) for _ in range(B)]
    a[0], b[0] = N - 1, N - 1
    A, Bm, Narr = ints_to_limbs(a, s2), ints_to_limbs(b, s2), int_to_limbs(N, s2)
    emu.set_group(group)
    try:
        emu.L.emu_set_mul_io(1)
        got = limbs_to_ints(emu.mulmod(Narr, A, Bm))
        assert emu.L.emu_last_mul_staged() == 1
        assert got == [x * y % N for x, y in zip(a, b)]
        emu.L.emu_set_mul_io(0)
        assert limbs_to_ints(emu.mulmod(Narr, A, Bm)) == got and emu.L.emu_last_mul_staged() == 0
        emu.L.emu_set_mul_io(1)
        out, rb = np.zeros_like(A), ctypes.c_int(0)
        assert emu.L.emu_montmul_rows(P(Narr), s2, P(A), P(Bm), 0, P(out), ctypes.c_uint64(B), ctypes.byref(rb)) == 0
        R = 1 << rb.value
        Rinv = pow(R, -1, N)
        assert limbs_to_ints(out) == [x * y * Rinv % N for x, y in zip(a, b)]
        for d in (0, 2):                                        # a debt of d factors of R, settled by the constant R^(d+1)
            debt = ints_to_limbs([x * pow(Rinv, d, N) % N for x in a], s2)
            const = ints_to_limbs([pow(R, d + 1, N)], s2)
            assert emu.L.emu_montmul_rows(P(Narr), s2, P(debt), P(const), 1, P(out), ctypes.c_uint64(B), None) == 0
            assert limbs_to_ints(out) == a
        m = [rng.randrange(n) for _ in range(B)]
        plain = emu.add_plain(int_to_limbs(n, s2 // 2), A, ints_to_limbs(m, s2 // 2))
        assert limbs_to_ints(plain) == [x * (1 + n * y) % N for x, y in zip(a, m)]
    finally:
        emu.set_group(0)
        emu.L.emu_set_mul_io(1)


def test_wide_key_products_on_the_pair_form(emu):
    """keys whose n^2 has no full-width geometry (above ~4170 bits; 8192 here, as in examples/benchmarks.py:88-90 of the
    reference): a*b mod n^2 and a*(1 + n*m) mod n^2 through split_core.h:mulmod_split_body"""
    g = load_golden(8192)
    n = H(g["n"])
    N = n * n
    rng = random.Random(1)
    a = [rng.randrange(N) for _ in range(3)] + [N - 1]
    b = [rng.randrange(N) for _ in range(3)] + [N - 1]
    n_arr = int_to_limbs(n, 256)
    got = emu.mulmod_n2_split(n_arr, ints_to_limbs(a, 512), ints_to_limbs(b, 512))
    assert limbs_to_ints(got) == [x * y % N for x, y in zip(a, b)]
    for e in g["raw_add"]:
        out = emu.mulmod_n2_split(n_arr, ints_to_limbs([H(e["a"])], 512), ints_to_limbs([H(e["b"])], 512))
        assert limbs_to_ints(out) == [H(e["out"])]
    m = [rng.randrange(n) for _ in range(4)]
    got = emu.mulmod_n2_split(n_arr, ints_to_limbs(a, 512), ints_to_limbs(m, 256), b_plain=True)
    assert limbs_to_ints(got) == [x * (1 + n * y) % N for x, y in zip(a, m)]


@pytest.mark.parametrize("key_bits,count", [(256, 20), (1024, 8), (2048, 4), (3072, 2)])
def test_key_owner_encryption_equals_the_golden_vectors(emu, key_bits, count):
    """raw_encrypt by the key owner (r^n from its CRT halves modulo p^2 and q^2, split_core.h:crt_lift_body) returns the
    ciphertexts the real reference returned, incl. m in {0, 1, max_int, ...} and r in {1, n-1}"""
    g = load_golden(key_bits)
    n = H(g["n"])
    s1 = key_bits // 32
    pq = max(1, s1 // 2)
    key = [int_to_limbs(H(g[k]), pq) for k in ("p", "q", "hp", "hq", "p_inverse")]
    enc = g["raw_encrypt"][:count]
    m = ints_to_limbs([H(e["m"]) % n for e in enc], s1)
    r = ints_to_limbs([H(e["r"]) for e in enc], s1)
    out = emu.encrypt_owner(int_to_limbs(n, s1), *key, m, r)
    assert out is not None and limbs_to_ints(out) == [H(e["c"]) for e in enc]


def test_key_owner_encryption_with_obfuscators_that_share_a_factor_with_n(emu):
    """r = p, q, multiples of them, 1, n - 1: r^n mod p^2 is 0 when p | r, the lift must still give pow(r, n, n^2)
    (the reference does not restrict r: phe/paillier.py:136-139)"""
    g = load_golden(256)
    n, p, q = H(g["n"]), H(g["p"]), H(g["q"])
    N = n * n
    key = [int_to_limbs(H(g[k]), 4) for k in ("p", "q", "hp", "hq", "p_inverse")]
    rng = random.Random(3)
    rs = [p, q, p * 3 % n, q * 5 % n, 1, n - 1, 2, p - 1, q + 1] + [rng.randrange(1, n) for _ in range(7)]
    ms = [0, n - 1, 1] + [rng.randrange(n) for _ in range(13)]
    out = emu.encrypt_owner(int_to_limbs(n, 8), *key, ints_to_limbs(ms, 8), ints_to_limbs(rs, 8))
    want = [(1 + n * m) * pow(r, n, N) % N for m, r in zip(ms, rs)]
    assert limbs_to_ints(out) == want
    assert limbs_to_ints(emu.encrypt(int_to_limbs(n, 8), ints_to_limbs(ms, 8), ints_to_limbs(rs, 8))) == want


@pytest.mark.parametrize("key_bits,count", [(256, None), (1024, 3), (2048, 2), (3072, 1)])
def test_one_number_on_a_wave_pair(emu, key_bits, count):
    """k_modexp_split_ab's body (split_core.h "one number on TWO wavefronts"): wave A runs the first words of every pair
    product, wave B — one product behind, through two LDS slots and one workgroup barrier per product — the second words.
    The two waves run in two host threads here; encrypt and both decrypt halves must give the golden bits."""
    emu.set_engine(True)
    emu.set_group(64)
    emu.set_unit(False)
    emu.set_wave_pairs(True)
    try:
        g = load_golden(key_bits)
        s1, s2, h = key_bits // 32, key_bits // 16, key_bits // 64
        n = int_to_limbs(H(g["n"]), s1)
        enc = g["raw_encrypt"]
        if count:
            enc = enc[:2] + enc[7:7 + count]                  # m = 0, 1 and a few random ones (r = 1 and n - 1 are further down)
        c = emu.encrypt(n, ints_to_limbs([H(e["m"]) for e in enc], s1), ints_to_limbs([H(e["r"]) for e in enc], s1))
        assert limbs_to_ints(c) == [H(e["c"]) for e in enc]
        dec = g["raw_decrypt"]
        if count:
            dec = dec[-count:]
        key = [int_to_limbs(H(g[k]), h) for k in ("p", "q", "hp", "hq", "p_inverse")]
        m = emu.decrypt(*key, s1, ints_to_limbs([H(e["c"]) for e in dec], s2))
        assert limbs_to_ints(m) == [H(e["m"]) for e in dec]
    finally:
        emu.set_wave_pairs(False)
        emu.set_unit(True)
        emu.set_group(0)


@pytest.mark.parametrize("group,key_bits,count", [(16, 256, None), (64, 256, None), (16, 1024, 3), (64, 1024, 2), (16, 2048, 3), (64, 2048, 2),
                                                  (16, 3072, 1), (64, 3072, 1)])
def test_small_batch_rungs_on_the_late_sweeps(emu, group, key_bits, count):
    """split_core.h pair_late / modexp_split_late_body (round 4): the single-wave kernels of the 16-lane and whole-wave rungs
    in the wave pair's row order — scaled modulus, quotient product after the shift, both words in one sweep with the first
    word one step ahead, rows = the limbs the scaled modulus needs (75 of 80 for a 2048-bit n on 16 x 5, 39 of 48 for its p on
    16 x 3), the way out modulo the true modulus.  encrypt (edge plaintexts m = 0, 1, ... and r = 1, n - 1 included at 256 bits),
    obfuscate and both decrypt halves must give the golden bits."""
    emu.set_engine(True)
    emu.set_group(group)
    emu.set_late(True)
    try:
        g = load_golden(key_bits)
        s1, s2, h = key_bits // 32, key_bits // 16, key_bits // 64
        n = int_to_limbs(H(g["n"]), s1)
        enc = g["raw_encrypt"]
        if count:
            enc = enc[:2] + enc[7:7 + count] + enc[-2:]
        c = emu.encrypt(n, ints_to_limbs([H(e["m"]) % H(g["n"]) for e in enc], s1), ints_to_limbs([H(e["r"]) for e in enc], s1))
        assert limbs_to_ints(c) == [H(e["c"]) for e in enc]
        dec = g["raw_decrypt"]
        if count:
            dec = dec[-count:] + dec[:1]
        key = [int_to_limbs(H(g[k]), h) for k in ("p", "q", "hp", "hq", "p_inverse")]
        m = emu.decrypt(*key, s1, ints_to_limbs([H(e["c"]) for e in dec], s2))
        assert limbs_to_ints(m) == [H(e["m"]) for e in dec]
        obf = g["obfuscate"][:count or None]
        out = emu.obfuscate(n, ints_to_limbs([H(e["c_in"]) for e in obf], s2), ints_to_limbs([H(e["r"]) for e in obf], s1))
        assert limbs_to_ints(out) == [H(e["c_out"]) for e in obf]
    finally:
        emu.set_late(False)
        emu.set_group(0)


def test_crt_halves_of_1024_bit_keys_run_one_lane_per_number(emu):
    """p, q of 512 bits are 18 limbs — ONE lane per number (G = 1, L = 18: no cross-lane step in any
    sweep, key_setup.h kS1) is rung 0 of the private side; the public side (n: 36 limbs) keeps 2 x 18.  Golden decrypts, junk
    ciphertexts that share a factor with n, and the key owner's encryption (same kernel, exponent n) give the reference's bits;
    wider keys are untouched."""
    emu.set_engine(True)
    emu.set_group(0)
    g = load_golden(1024)
    key = [int_to_limbs(H(g[k]), 16) for k in ("p", "q", "hp", "hq", "p_inverse")]
    assert emu.private_split_geometry(*key, 32) == (1, 18)
    emu.set_group(2)
    assert emu.private_split_geometry(*key, 32) == (2, 9)                    # the next rung of the ladder
    emu.set_group(0)
    g2 = load_golden(2048)
    key2 = [int_to_limbs(H(g2[k]), 32) for k in ("p", "q", "hp", "hq", "p_inverse")]
    assert emu.private_split_geometry(*key2, 64) == (2, 18)
    n, p, q = H(g["n"]), H(g["p"]), H(g["q"])
    dec = g["raw_decrypt"]
    junk = [0, p, 3 * p, q, n, 7 * n, p * p, n * n - p]
    cs = [H(e["c"]) for e in dec] + junk
    for wave_tail in (False, True):
        emu.set_wave_tail(wave_tail)
        try:
            m = emu.decrypt(*key, 32, ints_to_limbs(cs, 64))
        finally:
            emu.set_wave_tail(False)
        got = limbs_to_ints(m)
        assert got[:len(dec)] == [H(e["m"]) for e in dec]
        ref = lambda c: ((((pow(c, p - 1, p * p) - 1) // p) * H(g["hp"]) % p), (((pow(c, q - 1, q * q) - 1) // q) * H(g["hq"]) % q))
        want = [mp + ((mq - mp) * H(g["p_inverse"]) % q) * p for mp, mq in map(ref, junk)]
        assert got[len(dec):] == want
    enc = g["raw_encrypt"][:6]
    out = emu.encrypt_owner(int_to_limbs(n, 32), *key, ints_to_limbs([H(e["m"]) % n for e in enc], 32), ints_to_limbs([H(e["r"]) for e in enc], 32))
    assert out is not None and limbs_to_ints(out) == [H(e["c"]) for e in enc]


@pytest.mark.parametrize("key_bits", [1024, 2048])
def test_product_by_one_plain_product_and_one_table_fold(emu, key_bits):
    """csrc/mul_table.h (round 4): a*b mod n^2 — phe/util.py:53-64 mulmod, phe/paillier.py:705-719
    _raw_add — as ONE plain product and ONE fold of the limbs above n^2's width against the key's table W^(P+i) mod n^2, one
    double-precision quotient estimate, r = y - q n^2 < 3 n^2, conditional subtractions: no Montgomery factor to repair, half
    the multiply-adds of the two-product form.  Every golden raw_add vector, operands 0 / 1 / n^2 - 1 / all-ones rows (any
    value of the row's width is a legal operand: the result is the canonical residue of the product), random rows."""
    g = load_golden(key_bits)
    s2 = key_bits // 16
    n = H(g["n"])
    N = n * n
    rng = random.Random(key_bits + 7)
    top = (1 << (32 * s2)) - 1
    pairs = [(H(e["a"]), H(e["b"])) for e in g["raw_add"]]
    pairs += [(0, 5), (1, N - 1), (N - 1, N - 1), (top, top), (top, 1), (N, 7), (N + 1, N + 1), (1 << (32 * s2 - 1), 3)]
    pairs += [(rng.randrange(N), rng.randrange(N)) for _ in range(9)]
    pairs += [(rng.randrange(top), rng.randrange(top)) for _ in range(3)]
    pairs += [(rng.randrange(N), rng.randrange(N)) for _ in range(70)]     # > 32 rows: every limb group of the 8 emulated waves
    pairs += [(top, top), (N - 1, 2)]                                      # works on a second, third ... row (staging reused)
    a = ints_to_limbs([x for x, _ in pairs], s2)
    b = ints_to_limbs([y for _, y in pairs], s2)
    out = emu.mulmod_table(int_to_limbs(N, s2), a, b)
    assert out is not None
    assert limbs_to_ints(out) == [x * y % N for x, y in pairs]
    emu.L.emu_mad_count.restype = ctypes.c_uint64
    emu.L.emu_mad_count(1)
    emu.mulmod_table(int_to_limbs(N, s2), a[:8], b[:8])                      # two full waves' worth of groups
    table_mads = int(emu.L.emu_mad_count(1)) // 8
    emu.mulmod(int_to_limbs(N, s2), a[:8], b[:8])
    two_products = int(emu.L.emu_mad_count(1)) // 8
    assert table_mads * 1.9 < two_products, (table_mads, two_products)        # about half the multiply-adds per product


@pytest.mark.parametrize("key_bits", [1024, 2048, 3072])
def test_product_by_tiles_with_the_fold_on_one_element_per_lane(emu, key_bits):
    """csrc/mul_tile.h (round 4): the arithmetic of mul_table.h by tiles of 64 products per workgroup, ONE ELEMENT PER LANE: rows cut
    to 29-bit digits in registers, the product as column blocks (a digit of a times a sliding window of b), the fold against the
    column-block table (the scalar-path words), the settle on 16-lane groups; the workgroup's waves emulated as host threads joined
    by the barriers.  Same canonical residues on the golden raw_add vectors, the edge operands (all-ones rows: every column sum
    at its maximum), and batches that are not a multiple of the tile (ragged last tile, more tiles than workgroups, fewer)."""
    g = load_golden(key_bits)
    s2 = key_bits // 16
    n = H(g["n"])
    N = n * n
    rng = random.Random(key_bits + 11)
    top = (1 << (32 * s2)) - 1
    pairs = [(H(e["a"]), H(e["b"])) for e in g["raw_add"]]
    pairs += [(0, 5), (1, N - 1), (N - 1, N - 1), (top, top), (top, 1), (N, 7), (N + 1, N + 1), (1 << (32 * s2 - 1), 3)]
    pairs += [(rng.randrange(top), rng.randrange(top)) for _ in range(6)]
    pairs += [(rng.randrange(N), rng.randrange(N)) for _ in range(200 - len(pairs))]
    pairs += [(top, top), (N - 1, 2), (3, 0)]                              # 203 rows: 4 tiles, the last one with 11 live rows
    a = ints_to_limbs([x for x, _ in pairs], s2)
    b = ints_to_limbs([y for _, y in pairs], s2)
    want = [x * y % N for x, y in pairs]
    Nl = int_to_limbs(N, s2)
    out = emu.mulmod_table(Nl, a, b, tiles=True, blocks=2)                 # 2 workgroups x 2 tiles each
    assert out is not None
    assert limbs_to_ints(out) == want
    assert limbs_to_ints(emu.mulmod_table(Nl, a[:70], b[:70], tiles=True, blocks=3)) == want[:70]   # a workgroup without a tile
    assert limbs_to_ints(emu.mulmod_table(Nl, a[:5], b[:5], tiles=True, blocks=1)) == want[:5]
    in_lds = emu.mulmod_table(Nl, a, b)                                    # ... and the bits of mul_table.h where its table fits LDS
    assert (in_lds is None) == (key_bits == 3072) and (in_lds is None or np.array_equal(out, in_lds))


def test_product_by_tiles_on_eight_waves_for_1024_bit_keys(emu):
    """Round 5: n^2 of a 1024-bit key needs 72 columns (2048 bits + 38 of fold headroom) — 8 waves x 9, the
    shape TileShape<9, 8> (512 threads, 58 KB of LDS: two workgroups per CU), where the 16-wave kernel pads to 16 x 5 = 80.  P = 71 of
    the 72 columns are the modulus's own: ONE low fold digit, the quotient estimate one limb further down, the settle on 8-lane
    groups.  Golden raw_add vectors, edge operands (all-ones rows: every column sum at its maximum), ragged tiles; the same residues as
    the 16-wave shape and as Python integers; other moduli of the shape's range; rows too wide for 2 x 72 product digits are not offered."""
    g = load_golden(1024)
    s2 = 64
    n = H(g["n"])
    N = n * n
    rng = random.Random(1024 + 8)
    top = (1 << (32 * s2)) - 1
    pairs = [(H(e["a"]), H(e["b"])) for e in g["raw_add"]]
    pairs += [(0, 5), (1, N - 1), (N - 1, N - 1), (top, top), (top, 1), (N, 7), (N + 1, N + 1), (1 << (32 * s2 - 1), 3)]
    pairs += [(rng.randrange(top), rng.randrange(top)) for _ in range(6)]
    pairs += [(rng.randrange(N), rng.randrange(N)) for _ in range(200 - len(pairs))]
    pairs += [(top, top), (N - 1, 2), (3, 0)]                              # 203 rows: 4 tiles, the last one with 11 live rows
    a = ints_to_limbs([x for x, _ in pairs], s2)
    b = ints_to_limbs([y for _, y in pairs], s2)
    want = [x * y % N for x, y in pairs]
    Nl = int_to_limbs(N, s2)
    out = emu.mulmod_table(Nl, a, b, tiles=True, blocks=2, waves=8)
    assert out is not None and limbs_to_ints(out) == want
    assert limbs_to_ints(emu.mulmod_table(Nl, a[:70], b[:70], tiles=True, blocks=3, waves=8)) == want[:70]   # a workgroup without a tile
    assert limbs_to_ints(emu.mulmod_table(Nl, a[:5], b[:5], tiles=True, blocks=1, waves=8)) == want[:5]
    assert np.array_equal(out, emu.mulmod_table(Nl, a, b, tiles=True, blocks=2))       # the 16 x 5 shape: the same bits
    assert emu.L.emu_table_mul_offered(Nl.ctypes.data_as(ctypes.c_void_p), s2) == 6   # what the library takes: tiles, on 8 waves
    emu.L.emu_mad_count.restype = ctypes.c_uint64
    emu.L.emu_mad_count(1)
    # rows 64 .. 127 are residues below N, nothing special: the tile takes the settle's single candidate (mul_tile.h
    # tile_settle_blocks: the quotient estimate of every element keeps away from an integer) — the count of record
    emu.mulmod_table(Nl, a[64:128], b[64:128], tiles=True, blocks=1, waves=8)
    eight = int(emu.L.emu_mad_count(1)) // 64
    emu.mulmod_table(Nl, a[64:128], b[64:128], tiles=True, blocks=1)
    sixteen = int(emu.L.emu_mad_count(1)) // 64
    emu.mulmod(Nl, a[64:128], b[64:128])
    two_products = int(emu.L.emu_mad_count(1)) // 64
    assert eight == 11160 and eight * 1.12 < sixteen and eight * 1.8 < two_products, (eight, sixteen, two_products)
    # the first tile holds the edge operands (products below N: the estimate says 0): three candidates, 2 x 72 more multiply-adds each
    emu.mulmod_table(Nl, a[:64], b[:64], tiles=True, blocks=1, waves=8)
    assert int(emu.L.emu_mad_count(1)) // 64 == 11160 + 2 * 2 * 72
    # other widths of the shape: random moduli of 2048 and 2040 bits (rows of 64 words), one of 1700 bits (rows of 56 words)
    for bits in (2048, 2040, 1700):
        M = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
        limbs = -(-bits // 128) * 4
        full = (1 << (32 * limbs)) - 1
        prs = [(full, full), (M - 1, M - 1), (M, 3), (1, 0)] + [(rng.randrange(full), rng.randrange(full)) for _ in range(20)]
        got = emu.mulmod_table(int_to_limbs(M, limbs), ints_to_limbs([x for x, _ in prs], limbs), ints_to_limbs([y for _, y in prs], limbs),
                               tiles=True, blocks=1, waves=8)
        assert got is not None and limbs_to_ints(got) == [x * y % M for x, y in prs], bits
    M = rng.getrandbits(2049) | (1 << 2048) | 1                            # rows of 68 words: their product has more than 2 x 72 digits
    assert emu.mulmod_table(int_to_limbs(M, 68), ints_to_limbs([5], 68), ints_to_limbs([6], 68), tiles=True, blocks=1, waves=8) is None


@pytest.mark.parametrize("bits", [3685, 3713, 3900, 4095, 4130, 5850, 6000, 6143])
def test_tile_product_on_moduli_of_every_width_a_lane_count_takes(emu, bits):
    """mul_tile.h / mul_table.h take ANY odd modulus whose limbs fill the lanes to within 16 (n_lo = S - P fold digits come from the
    low half): widths across the range of L = 9 (S = 144) and L = 14 (S = 224), first and last digit counts included; the same
    residues as Python integers for operands up to the full row width.  Widths between the lane counts are not offered."""
    rng = random.Random(bits)
    N = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
    limbs = -(-bits // 128) * 4                                            # rows of whole 16-byte pieces
    top = (1 << (32 * limbs)) - 1
    pairs = [(top, top), (N - 1, N - 1), (N, 3), (1, 0)] + [(rng.randrange(top), rng.randrange(top)) for _ in range(6)]
    pairs += [(rng.randrange(N), rng.randrange(N)) for _ in range(60)]
    a = ints_to_limbs([x for x, _ in pairs], limbs)
    b = ints_to_limbs([y for _, y in pairs], limbs)
    out = emu.mulmod_table(int_to_limbs(N, limbs), a, b, tiles=True, blocks=2)
    digits = -(-bits // 29)
    lanes = 144 if digits <= 144 else 224
    offered = 2 <= lanes - digits <= 16 and -(-(bits + 38) // 29) <= lanes
    assert (out is not None) == offered, (bits, digits)
    if offered:
        assert limbs_to_ints(out) == [x * y % N for x, y in pairs]


def test_table_product_is_not_offered_where_the_table_does_not_fit(emu):
    g = load_golden(3072)
    N = H(g["n"]) ** 2
    a = ints_to_limbs([5, 6], 192)
    assert emu.mulmod_table(int_to_limbs(N, 192), a, a) is None                        # 212 rows of 224 limbs: not in LDS
    assert limbs_to_ints(emu.mulmod_table(int_to_limbs(N, 192), a, a, tiles=True)) == [25, 36]   # ... the tile kernel keeps it in L2
    p4 = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "paillier_4096_primes.json")))
    N = (int(p4["p"], 16) * int(p4["q"], 16)) ** 2
    a = ints_to_limbs([5, 6], 256)
    assert emu.mulmod_table(int_to_limbs(N, 256), a, a, tiles=True) is None              # 4096-bit keys: no lane width compiled


@pytest.mark.parametrize("key_bits", [256, 1024, 2048, 3072])
def test_decrypt_tail_on_one_wave_per_ciphertext(emu, key_bits):
    """split_core.h decrypt_tail_wave_body (the tail the library takes for small batches): L-function, * hp, CRT on the
    whole-wave sweeps; same plaintexts as the per-thread tail on every golden raw_decrypt vector, incl. m = 0, 1, n - 1."""
    emu.set_engine(True)
    emu.set_wave_tail(True)
    try:
        g = load_golden(key_bits)
        s1, s2, h = key_bits // 32, key_bits // 16, key_bits // 64
        dec = g["raw_decrypt"] if key_bits <= 1024 else g["raw_decrypt"][:6]
        key = [int_to_limbs(H(g[k]), h) for k in ("p", "q", "hp", "hq", "p_inverse")]
        n = H(g["n"])
        cts = [H(e["c"]) for e in dec]
        want = [H(e["m"]) for e in dec]
        N = n * n
        for m in (0, 1, n - 1, n - 2):                                  # edge plaintexts: c = (1 + n*m) * 2^n
            cts.append((1 + n * m) * pow(2, n, N) % N)
            want.append(m)
        # "ciphertexts" that share a factor with n (c^(p-1) mod p^2 = 0): the reference's l_function floors (0 - 1) // p to -1
        p, q = H(g["p"]), H(g["q"])

        def reference_value(c):                                         # phe/paillier.py:346-374 on Python integers
            mp = (pow(c, p - 1, p * p) - 1) // p * H(g["hp"]) % p
            mq = (pow(c, q - 1, q * q) - 1) // q * H(g["hq"]) % q
            return mp + (mq - mp) * H(g["p_inverse"]) % q * p
        for c in (0, p, 3 * p, q, n, 7 * n, p * p, N - p):
            cts.append(c)
            want.append(reference_value(c))
        got = limbs_to_ints(emu.decrypt(*key, s1, ints_to_limbs(cts, s2)))
        assert got == want
        emu.set_wave_tail(False)                                        # ... and the one-ciphertext-per-thread tail agrees
        assert limbs_to_ints(emu.decrypt(*key, s1, ints_to_limbs(cts[-8:], s2))) == want[-8:]
    finally:
        emu.set_wave_tail(False)


def test_wave_pair_sweeps_with_three_limbs_per_lane(emu):
    """A 4096-bit key (primes of tests/golden/paillier_4096_primes.json, made by gen_primes_4096.py): n~ needs 143 limbs = three per
    lane, the width at which the first word's columns still carry 32-bit carries and the second word's (three products per step
    in a multiplication) no longer do (split_core.h ab_shift_narrow); its CRT halves run two limbs per lane.  One encryption and
    its decryption on wave pairs, the tail on one wave, against CPython's pow."""
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "paillier_4096_primes.json")) as f:
        pq = json.load(f)
    p, q = int(pq["p"], 16), int(pq["q"], 16)
    rng = random.Random(4096)
    n = p * q
    N = n * n
    hp = pow((pow(n + 1, p - 1, p * p) - 1) // p, -1, p)                # phe/paillier.py:234-235, h_function
    hq = pow((pow(n + 1, q - 1, q * q) - 1) // q, -1, q)
    key = [int_to_limbs(v, 64) for v in (p, q, hp, hq, pow(p, -1, q))]
    m, r = rng.randrange(n), rng.randrange(1, n)
    emu.set_engine(True)
    emu.set_group(64)
    emu.set_unit(False)
    emu.set_wave_pairs(True)
    emu.set_wave_tail(True)
    try:
        c = emu.encrypt(int_to_limbs(n, 128), ints_to_limbs([m], 128), ints_to_limbs([r], 128))
        assert limbs_to_ints(c) == [(1 + n * m) * pow(r, n, N) % N]
        assert limbs_to_ints(emu.decrypt(*key, 128, c)) == [m]
    finally:
        emu.set_wave_tail(False)
        emu.set_wave_pairs(False)
        emu.set_unit(True)
        emu.set_group(0)


@pytest.mark.parametrize("key_bits", [1600, 2100, 2240])
def test_wave_pair_sweeps_on_key_sizes_off_the_grid(emu, key_bits):
    """Key sizes whose limb counts are not round (primes of tests/golden/paillier_odd_sizes_primes.json): the sweeps of a wave
    pair end in a trip of 1, 2, 3 or 4 steps (a compile-time variant each, split_core.h ab_first_word) — the golden key sizes
    hit 2, 3 and 4 for the CRT halves, these hit 1 (2240 bits, also for n on two limbs per lane), 2 (1600) and 3 (2100).
    One encryption and its decryption (tail on one wave), plus m = 0 and m = n - 1, against CPython's pow."""
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "paillier_odd_sizes_primes.json")) as f:
        pq = json.load(f)[str(key_bits)]
    p, q = int(pq["p"], 16), int(pq["q"], 16)
    n = p * q
    N = n * n
    s1 = 2 * ((key_bits + 63) // 64)                                    # limbs of n: even, so that p and q take half each
    hp = pow((pow(n + 1, p - 1, p * p) - 1) // p, -1, p)                # phe/paillier.py:234-235, h_function
    hq = pow((pow(n + 1, q - 1, q * q) - 1) // q, -1, q)
    key = [int_to_limbs(v, s1 // 2) for v in (p, q, hp, hq, pow(p, -1, q))]
    rng = random.Random(key_bits)
    ms = [rng.randrange(n), 0, n - 1]
    rs = [rng.randrange(1, n), n - 1, rng.randrange(1, n)]
    emu.set_engine(True)
    emu.set_group(64)
    emu.set_unit(False)
    emu.set_wave_pairs(True)
    emu.set_wave_tail(True)
    try:
        c = emu.encrypt(int_to_limbs(n, s1), ints_to_limbs(ms, s1), ints_to_limbs(rs, s1))
        assert limbs_to_ints(c) == [(1 + n * m) * pow(r, n, N) % N for m, r in zip(ms, rs)]
        assert limbs_to_ints(emu.decrypt(*key, s1, c)) == ms
    finally:
        emu.set_wave_tail(False)
        emu.set_wave_pairs(False)
        emu.set_unit(True)
        emu.set_group(0)


@pytest.mark.parametrize("key_bits", [256, 1024, 2048])
def test_scalar_multiplication_of_a_handful_on_wave_pairs(emu, key_bits):
    """phe_hip_powmod for a handful of numbers (EncryptedNumber.__mul__, one at a time): every number on a wave pair with its own
    sliding-window schedule — the golden _raw_mul vectors of the direct branch and 1-, 56-, 64-bit and full-width exponents, against
    CPython's pow; an exponent 0 in the batch keeps the general kernel."""
    emu.set_engine(True)
    emu.set_group(64)
    emu.set_wave_pairs(True)
    try:
        g = load_golden(key_bits)
        n = H(g["n"])
        N, s1, s2 = n * n, key_bits // 32, key_bits // 16
        rng = random.Random(key_bits + 11)
        direct = [e for e in g["raw_mul"] if 0 < H(e["s"]) < n - H(g["max_int"])][:3]
        cs = [H(e["c"]) for e in direct] + [rng.randrange(1, N) for _ in range(4)]
        ks = [H(e["s"]) for e in direct] + [1, (1 << 56) - 1, rng.getrandbits(64), rng.randrange(n // 3)]
        got = limbs_to_ints(emu.powmod_n2(int_to_limbs(n, s1), ints_to_limbs(cs, s2), ints_to_limbs(ks, s1)))
        assert got == [pow(c, k, N) for c, k in zip(cs, ks)]
        assert got[:len(direct)] == [H(e["out"]) for e in direct]
        got = limbs_to_ints(emu.powmod_n2(int_to_limbs(n, s1), ints_to_limbs(cs[:2], s2), ints_to_limbs([0, 5], 1)))
        assert got == [1, pow(cs[1], 5, N)]
    finally:
        emu.set_wave_pairs(False)
        emu.set_group(0)


def test_split_geometries_by_modulus_width(emu):
    """pick_geometry_split (key_setup.h) over the compiled limb counts: the rungs a modulus of a given width gets when a group
    width is preferred — p, q of 1024-bit keys (18 limbs) have a 4-lane and an 8-lane rung of their own (4 x 5, 8 x 3), n of a
    1024-bit key (36 limbs) a 4-lane one (4 x 9); the 2048- and 3072-bit geometries are what the GPU sweeps were measured on."""
    def geom(bits, group):
        emu.set_group(group)
        try:
            return emu.split_geometry(int_to_limbs((1 << bits) - 159, (bits + 31) // 32))
        finally:
            emu.set_group(0)
    want = {
        512: {1: (1, 18), 0: (2, 9), 4: (4, 5), 8: (8, 3), 16: (16, 2), 64: (64, 1)},   # p, q of a 1024-bit key (1: one lane per
                                                                                 # number, what the private side's rung 0 asks for)
        1023: {1: (2, 18)},                                                      # no one-lane geometry above 18 limbs
        1024: {0: (2, 18), 4: (4, 9), 8: (8, 5), 16: (16, 3), 64: (64, 1)},      # n of a 1024-bit key; p, q of a 2048-bit key
        2048: {0: (4, 18), 4: (4, 18), 8: (8, 9), 16: (16, 5), 64: (64, 2)},     # n of a 2048-bit key
        1536: {0: (2, 27), 4: (4, 14), 8: (8, 7), 16: (16, 4), 64: (64, 1)},     # p, q of a 3072-bit key
        3072: {0: (4, 27), 8: (8, 14), 16: (16, 7), 64: (64, 2)},                # n of a 3072-bit key
    }
    for bits, by_group in want.items():
        for group, gl in by_group.items():
            assert geom(bits, group) == gl, (bits, group)
