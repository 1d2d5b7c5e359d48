"""Pin the CPU oracle (oracle/) before anything trusts it.

(a) the reference's own known-answer vectors (phe/tests/paillier_test.py:128-149,
    phe/tests/util_test.py:29-44), restated in tests/golden/reference_kat.json;
(b) tests/golden/paillier_{256,1024,2048,3072}.json, produced by tests/golden/gen_golden.py
    from the real reference imported in the build container.
Both the pure-Python restatement and the C/libgmp restatement must reproduce every value.
"""
import numpy as np
import pytest

from conftest import load_golden, load_kat
from oracle.paillier_oracle import (PyPrivate, PyPublic, int_to_limbs, ints_to_limbs, limbs_to_ints,
                                    py_invert, py_powmod)

KEYS = [256, 1024, 2048, 3072]


def H(x):
    return int(x, 16)


def test_reference_kat_python():
    k = load_kat()
    pub = PyPublic(k["n"])
    priv = PyPrivate(pub, k["p"], k["q"])
    assert pub.raw_encrypt(k["m"], k["r"]) == k["c"]
    assert priv.raw_decrypt(k["c"]) == k["m"]
    assert pub.raw_encrypt(1, 1) == k["encrypt_1_r_1"]
    assert (priv.hp, priv.hq, priv.p_inverse) == (k["hp"], k["hq"], k["p_inverse"])
    for a, b, c, want in k["powmod"]:
        assert py_powmod(a, b, c) == want
    for a, b, want in k["invert"]:
        assert py_invert(a, b) == want


def test_reference_kat_c(c_oracle):
    k = load_kat()
    n = int_to_limbs(k["n"], 1)
    c = c_oracle.encrypt(n, ints_to_limbs([k["m"], 1], 1), ints_to_limbs([k["r"], 1], 1))
    assert limbs_to_ints(c) == [k["c"], k["encrypt_1_r_1"]]
    m = c_oracle.decrypt(n, int_to_limbs(k["q"], 1), int_to_limbs(k["p"], 1), c[:1])  # unordered p,q
    assert limbs_to_ints(m) == [k["m"]]
    assert c_oracle.private_constants(k["n"], k["q"], k["p"], 1, 1) == (
        k["p"], k["q"], k["hp"], k["hq"], k["p_inverse"])
    for a, b, cc, want in k["powmod"]:
        assert c_oracle.powmod(a, b, cc) == want
    for a, b, want in k["invert"]:
        assert c_oracle.invert(a, b) == want
    with pytest.raises(ZeroDivisionError):
        c_oracle.invert(2, 4)


@pytest.mark.parametrize("key_bits", KEYS)
def test_golden_python(key_bits):
    g = load_golden(key_bits)
    if key_bits > 1024:  # CPython pow at 2048+ bits is slow; sample
        take = lambda xs: xs[::3]
    else:
        take = lambda xs: xs
    pub = PyPublic(H(g["n"]))
    priv = PyPrivate(pub, H(g["q"]), H(g["p"]))
    assert (priv.hp, priv.hq, priv.p_inverse) == (H(g["hp"]), H(g["hq"]), H(g["p_inverse"]))
    for e in take(g["raw_encrypt"]):
        assert pub.raw_encrypt(H(e["m"]), H(e["r"])) == H(e["c"])
    for e in take(g["raw_decrypt"]):
        assert priv.raw_decrypt(H(e["c"])) == H(e["m"])
    for e in g["obfuscate"]:
        assert pub.obfuscate(H(e["c_in"]), H(e["r"])) == H(e["c_out"])
    for e in g["raw_add"]:
        assert pub.raw_add(H(e["a"]), H(e["b"])) == H(e["out"])
    for e in take(g["raw_mul"]):
        assert pub.raw_mul(H(e["c"]), H(e["s"])) == H(e["out"])


@pytest.mark.parametrize("key_bits", KEYS + [8192])
def test_golden_c(c_oracle, key_bits):
    g = load_golden(key_bits)
    s1 = key_bits // 32
    s2 = 2 * s1
    n = int_to_limbs(H(g["n"]), s1)
    p = int_to_limbs(H(g["p"]), s1 // 2)
    q = int_to_limbs(H(g["q"]), s1 // 2)
    assert c_oracle.private_constants(H(g["n"]), H(g["q"]), H(g["p"]), s1, s1 // 2) == (
        H(g["p"]), H(g["q"]), H(g["hp"]), H(g["hq"]), H(g["p_inverse"]))

    enc = g["raw_encrypt"]
    # the C-ABI carries m in s1 limbs; the fixture also holds m = n and n+1 (fit) — all < 2^(32*s1)
    m = ints_to_limbs([H(e["m"]) for e in enc], s1)
    r = ints_to_limbs([H(e["r"]) for e in enc], s1)
    c = c_oracle.encrypt(n, m, r, nthreads=4)
    assert limbs_to_ints(c) == [H(e["c"]) for e in enc]

    dec = g["raw_decrypt"]
    cin = ints_to_limbs([H(e["c"]) for e in dec], s2)
    assert limbs_to_ints(c_oracle.decrypt(n, q, p, cin, nthreads=4)) == [H(e["m"]) for e in dec]

    obf = g["obfuscate"]
    got = c_oracle.obfuscate(n, ints_to_limbs([H(e["c_in"]) for e in obf], s2),
                             ints_to_limbs([H(e["r"]) for e in obf], s1))
    assert limbs_to_ints(got) == [H(e["c_out"]) for e in obf]

    add = g["raw_add"]
    got = c_oracle.add(n, ints_to_limbs([H(e["a"]) for e in add], s2), ints_to_limbs([H(e["b"]) for e in add], s2))
    assert limbs_to_ints(got) == [H(e["out"]) for e in add]

    mul = g["raw_mul"]
    got = c_oracle.mul(n, ints_to_limbs([H(e["c"]) for e in mul], s2), ints_to_limbs([H(e["s"]) for e in mul], s1),
                       nthreads=4)
    assert limbs_to_ints(got) == [H(e["out"]) for e in mul]


def test_c_oracle_error_conventions(c_oracle):
    g = load_golden(256)
    s1 = 8
    n_int = H(g["n"])
    n = int_to_limbs(n_int, s1)
    c = ints_to_limbs([H(g["raw_encrypt"][0]["c"])], 2 * s1)
    with pytest.raises(ValueError):  # phe/paillier.py:742-743
        c_oracle.mul(n, c, ints_to_limbs([n_int], s1))
    # non-invertible ciphertext (multiple of p) on the inverse branch -> ZeroDivisionError (phe/util.py:96-97)
    bad = ints_to_limbs([H(g["p"])], 2 * s1)
    with pytest.raises(ZeroDivisionError):
        c_oracle.mul(n, bad, ints_to_limbs([n_int - 1], s1))
