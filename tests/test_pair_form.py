"""Resident rows in the pair form (csrc/split_core.h to_pair_body / from_pair_body / pair_mul_body; include/phe_hip.h
"pair form") on the CPU wave emulator: the bodies the GPU runs, compiled for the host.  What leaves the pair form must be
the canonical residue the reference's chain of _raw_add (phe/paillier.py:705-719: mulmod(a, b, nsquare)) / raw_encrypt
(:134-139) returns — checked against Python integers on the golden keys made by the real reference."""
import random

import numpy as np
import pytest

from conftest import load_golden
from emu_lib import Emu

H = lambda s: int(s, 16)


@pytest.fixture(scope="module")
def emu():
    return Emu()


def limbs(xs, words):
    raw = b"".join(int(x).to_bytes(4 * words, "little") for x in xs)
    return np.frombuffer(raw, np.uint32).reshape(len(xs), words).copy()


def ints(arr):
    w = arr.shape[1] * 4
    raw = arr.tobytes()
    return [int.from_bytes(raw[i * w:(i + 1) * w], "little") for i in range(arr.shape[0])]


@pytest.mark.parametrize("bits", [256, 1024, 2048])
def test_round_trip_products_and_plaintext_factor(emu, bits):
    g = load_golden(bits)
    n = H(g["n"])
    n2, s1 = n * n, (n.bit_length() + 31) // 32
    n_arr = limbs([n], s1)[0]
    words = emu.pair_words(n_arr)
    assert words and words % 2 == 0 and 29 * (words // 2) >= n.bit_length() + 4
    rng = random.Random(bits)
    rows = 5 if bits == 2048 else 9
    # edge residues next to random ones: 0, 1, n, n^2 - 1, and a value above n^2 that still fits the words
    a = [0, 1, n, n2 - 1, (1 << (64 * s1)) - 1][:rows] + [rng.randrange(n2) for _ in range(max(0, rows - 5))]
    b = [rng.randrange(n2) for _ in range(rows)]
    pa, pb = emu.pair_op(n_arr, 0, limbs(a, 2 * s1)), emu.pair_op(n_arr, 0, limbs(b, 2 * s1))
    assert pa.shape == (rows, words) and int(pa.max()) < (1 << 29) + (1 << 8)          # almost-normalised 29-bit limbs
    assert ints(emu.pair_op(n_arr, 1, pa)) == [x % n2 for x in a]                       # out again: the canonical residue
    prod = emu.pair_op(n_arr, 2, pa, pb)
    assert ints(emu.pair_op(n_arr, 1, prod)) == [x * y % n2 for x, y in zip(a, b)]      # _raw_add
    chain = prod
    want = [x * y % n2 for x, y in zip(a, b)]
    for _ in range(3):                                                                   # a chain stays in the pair form
        chain = emu.pair_op(n_arr, 2, chain, pb)
        want = [w * y % n2 for w, y in zip(want, b)]
    assert ints(emu.pair_op(n_arr, 1, chain)) == want
    row = emu.pair_op(n_arr, 2, pa, pb[:1], b_is_row=True)                               # one row b for every a
    assert ints(emu.pair_op(n_arr, 1, row)) == [x * b[0] % n2 for x in a]
    m = [0, 1, n - 1, n, (1 << (32 * s1)) - 1][:rows] + [rng.randrange(n) for _ in range(max(0, rows - 5))]
    got = ints(emu.pair_op(n_arr, 1, pb, limbs(m, s1)))                                  # x * (1 + n*m): phe/paillier.py:134-139
    assert got == [y * (1 + n * (mm % n)) % n2 for y, mm in zip(b, m)]


def test_wider_rung_serves_the_same_rows(emu):
    """2048-bit keys: rung 0 is groups of 4 lanes x 18 limbs, the next rung 8 x 9 — the same 72 limbs, so small batches of
    pair rows run there (phe_hip.hip:pick_pair_split); a 16-lane rung (80 limbs) cannot take them"""
    g = load_golden(2048)
    n = H(g["n"])
    n2, s1 = n * n, 64
    n_arr = limbs([n], s1)[0]
    rng = random.Random(7)
    a, b = [rng.randrange(n2) for _ in range(3)], [rng.randrange(n2) for _ in range(3)]
    pa = emu.pair_op(n_arr, 0, limbs(a, 128))
    pb8 = emu.pair_op(n_arr, 0, limbs(b, 128), group=8)
    assert pb8 is not None
    prod8 = emu.pair_op(n_arr, 2, pa, pb8, group=8)
    assert ints(emu.pair_op(n_arr, 1, prod8, group=8)) == [x * y % n2 for x, y in zip(a, b)]
    assert ints(emu.pair_op(n_arr, 1, prod8)) == [x * y % n2 for x, y in zip(a, b)]      # and back on rung 0
    assert emu.pair_op(n_arr, 0, limbs(a, 128), group=16) is None


def test_golden_raw_add_through_the_pair_form(emu):
    """the reference's own _raw_add vectors (tests/golden, made by the imported reference)"""
    g = load_golden(1024)
    n = H(g["n"])
    s1 = (n.bit_length() + 31) // 32
    n_arr = limbs([n], s1)[0]
    adds = g["raw_add"]
    pa = emu.pair_op(n_arr, 0, limbs([H(e["a"]) for e in adds], 2 * s1))
    pb = emu.pair_op(n_arr, 0, limbs([H(e["b"]) for e in adds], 2 * s1))
    assert ints(emu.pair_op(n_arr, 1, emu.pair_op(n_arr, 2, pa, pb))) == [H(e["out"]) for e in adds]
    enc = g["raw_encrypt"]
    # raw_encrypt(m, r) = (1 + n*m) * r^n: r^n in the pair form (as the obfuscator pool keeps it), the plaintext folded in on the way out
    rn = [pow(H(e["r"]), n, n * n) for e in enc]
    got = ints(emu.pair_op(n_arr, 1, emu.pair_op(n_arr, 0, limbs(rn, 2 * s1)), limbs([H(e["m"]) % (1 << (32 * s1)) for e in enc], s1)))
    assert got == [H(e["c"]) for e in enc]


@pytest.mark.parametrize("bits", [256, 1024, 2048])
def test_scalar_multiplication_stays_in_the_pair_form(emu, bits):
    """modexp_var_split_body<G, L, PAIR> (phe_hip_pair_powmod_dev): a[i]^k[i] on pair rows, pair rows out — _raw_mul by
    non-negative scalars (phe/paillier.py:721-751: powmod(c, k, nsquare)) without the conversion in and the exit.  The golden
    raw_mul vectors of the reference whose scalar takes the direct branch, then random ciphertexts with scalars 0, 1, 2, 2^56 - 1
    and 64-bit ones; on the wider rung that shares the rows' limb count as well."""
    g = load_golden(bits)
    n = H(g["n"])
    n2, s1 = n * n, (n.bit_length() + 31) // 32
    n_arr = limbs([n], s1)[0]
    rng = random.Random(bits + 5)
    cs = [rng.randrange(1, n2) for _ in range(7)]
    ks = [0, 1, 2, (1 << 56) - 1, rng.getrandbits(64), rng.getrandbits(53), 3]
    want = [pow(c, k, n2) for c, k in zip(cs, ks)]
    # the reference's own _raw_mul vectors whose scalar takes the direct branch (s < n - max_int: powmod(c, s, nsquare))
    direct = [e for e in g["raw_mul"] if H(e["s"]) < n - H(g["max_int"])][:6]
    assert direct
    pg = emu.pair_op(n_arr, 0, limbs([H(e["c"]) for e in direct], 2 * s1))
    outg = emu.pair_powmod(n_arr, pg, limbs([H(e["s"]) for e in direct], s1))
    assert ints(emu.pair_op(n_arr, 1, outg)) == [H(e["out"]) for e in direct]
    pa = emu.pair_op(n_arr, 0, limbs(cs, 2 * s1))
    out = emu.pair_powmod(n_arr, pa, limbs(ks, 2))
    assert out is not None and ints(emu.pair_op(n_arr, 1, out)) == want
    # a chain that never leaves the form: (c^k) * c2, then ^3
    pb = emu.pair_op(n_arr, 0, limbs(cs[::-1], 2 * s1))
    chain = emu.pair_powmod(n_arr, emu.pair_op(n_arr, 2, out, pb), limbs([3] * len(cs), 1))
    assert ints(emu.pair_op(n_arr, 1, chain)) == [pow(w * c2 % n2, 3, n2) for w, c2 in zip(want, cs[::-1])]
    if bits == 2048:
        out8 = emu.pair_powmod(n_arr, pa, limbs(ks, 2), group=8)
        assert out8 is not None and ints(emu.pair_op(n_arr, 1, out8)) == want
