"""CPU tests of the code the GPU runs: csrc/mont_core.h, csrc/decrypt_tail.h and csrc/key_setup.h are
compiled for the host on a fiber-based wavefront emulator (tests/emu/) and compared, limb for limb,
with the golden fixtures (real reference) and the libgmp oracle."""
import random

import numpy as np
import pytest

from conftest import load_golden, load_kat
from emu_lib import Emu
from oracle.paillier_oracle import int_to_limbs, ints_to_limbs, limbs_to_int, limbs_to_ints


def H(x):
    return int(x, 16)


@pytest.fixture(scope="module")
def emu():
    return Emu()


@pytest.mark.parametrize("L", [1, 2, 3, 4, 6, 8, 12, 16])
def test_montmul_rows(emu, L):
    rng = random.Random(100 + L)
    S = 16 * L
    R = 1 << (32 * S)
    moduli = [rng.getrandbits(32 * S) | 1 | (1 << (32 * S - 1)),  # full width
              (R - 1 - 2 * rng.getrandbits(20)) | 1,             # just below R: exercises the overflow bit
              126869 ** 2,                                        # tiny modulus in a wide container
              (1 << (32 * S - 1)) + 1]
    for N in moduli:
        a = [rng.randrange(0, R) for _ in range(4)]
        b = [rng.randrange(0, N) for _ in range(4)]
        a[0], b[0], a[1], b[2] = R - 1, N - 1, 0, 0
        got = limbs_to_ints(emu.montmul(L, ints_to_limbs(a, S), ints_to_limbs(b, S), int_to_limbs(N, S)))
        rinv = pow(R, -1, N)
        assert got == [x * y * rinv % N for x, y in zip(a, b)]


def test_carry_chains_all_ones(emu):
    # products engineered so lane sums are all-ones words and carries ripple across every lane
    L, S = 2, 32
    R = 1 << (32 * S)
    N = R - 1  # odd, every limb 0xffffffff
    a = [R - 1, R - 2, 1, (1 << 512) - 1]
    b = [N - 1, N - 1, N - 1, N - 2]
    got = limbs_to_ints(emu.montmul(L, ints_to_limbs(a, S), ints_to_limbs(b, S), int_to_limbs(N, S)))
    rinv = pow(R, -1, N)
    assert got == [x * y * rinv % N for x, y in zip(a, b)]


@pytest.mark.parametrize("key_bits", [256, 1024])
def test_public_constants(emu, key_bits):
    g = load_golden(key_bits)
    n = H(g["n"])
    s1 = key_bits // 32
    k = emu.public_constants(int_to_limbs(n, s1))
    S = k["S"]
    N = n * n
    R = 1 << (32 * S)
    assert S >= 2 * s1 and S == 16 * k["L"]
    assert limbs_to_int(k["n"]) == N
    assert limbs_to_int(k["r1"]) == R % N
    assert limbs_to_int(k["r2"]) == R * R % N
    assert limbs_to_int(k["r3"]) == R * R * R % N
    assert limbs_to_int(k["aux"]) == n * R % N
    assert (k["n0inv"] * N + 1) % (1 << 32) == 0
    # schedule cost vs the canonical E(t) = t + t/6 + 16 of SURVEY.md 8(d)
    assert k["squarings"] <= key_bits and k["multiplies"] <= key_bits // 5 + 2


def test_reference_kat(emu):
    k = load_kat()
    n = int_to_limbs(k["n"], 1)
    c = emu.encrypt(n, ints_to_limbs([k["m"], 1], 1), ints_to_limbs([k["r"], 1], 1))
    assert limbs_to_ints(c) == [k["c"], k["encrypt_1_r_1"]]
    one = lambda x: int_to_limbs(x, 1)
    m = emu.decrypt(one(k["p"]), one(k["q"]), one(k["hp"]), one(k["hq"]), one(k["p_inverse"]), 1, c)
    assert limbs_to_ints(m) == [k["m"], 1]


@pytest.mark.parametrize("key_bits,count", [(256, None), (1024, 5), (2048, 2)])
def test_golden_through_emulator(emu, key_bits, count):
    g = load_golden(key_bits)
    s1, s2, h = key_bits // 32, key_bits // 16, key_bits // 64
    n = int_to_limbs(H(g["n"]), s1)
    enc = g["raw_encrypt"]
    if count:  # edge cases first (m = 0, 1, ..., n-1, n, n+1), then a few random
        enc = enc[8:8 + count] if key_bits > 1024 else enc[7:7 + count]
    c = emu.encrypt(n, ints_to_limbs([H(e["m"]) for e in enc], s1), ints_to_limbs([H(e["r"]) for e in enc], s1))
    assert limbs_to_ints(c) == [H(e["c"]) for e in enc]

    dec = g["raw_decrypt"]
    if count:
        dec = dec[-count:]
    key = [int_to_limbs(H(g[k]), h) for k in ("p", "q", "hp", "hq", "p_inverse")]
    m = emu.decrypt(*key, s1, ints_to_limbs([H(e["c"]) for e in dec], s2))
    assert limbs_to_ints(m) == [H(e["m"]) for e in dec]


def test_homomorphic_ops_through_emulator(emu):
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
    got = emu.powmod_var(N, ints_to_limbs([H(e["c"]) for e in mul], s2), ints_to_limbs([H(e["s"]) for e in mul], s1))
    assert limbs_to_ints(got) == [H(e["out"]) for e in mul]
    # negative branch: powmod(invert(c), n - s) (phe/paillier.py:745-749); the inverse comes from Python here
    neg = [e for e in g["raw_mul"] if H(e["s"]) >= n_int - max_int]
    bases = [pow(H(e["c"]), -1, n_int * n_int) for e in neg]
    got = emu.powmod_var(N, ints_to_limbs(bases, s2), ints_to_limbs([n_int - H(e["s"]) for e in neg], s1))
    assert limbs_to_ints(got) == [H(e["out"]) for e in neg]


def test_powmod_var_mixed_lengths(emu):
    rng = random.Random(7)
    S = 16
    N_int = (rng.getrandbits(500) | 1 | (1 << 499))
    N = int_to_limbs(N_int, S)
    bases = [rng.randrange(1, N_int) for _ in range(7)]
    exps = [0, 1, 2, 16 ** 7, rng.getrandbits(56), rng.getrandbits(200), (1 << 64) - 1]
    got = emu.powmod_var(N, ints_to_limbs(bases, S), ints_to_limbs(exps, 8))
    assert limbs_to_ints(got) == [pow(b, e, N_int) for b, e in zip(bases, exps)]
