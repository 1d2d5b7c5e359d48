"""Array forms of the codec and of the random draw (python-paillier_amd/phe/codec.py, _engine.py) against their
scalar definitions — EncodedNumber.encode / decode (the mirror of phe/encoding.py:110-233) and
PaillierPublicKey.get_random_lt_n (phe/paillier.py:141-143).  Host-only."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import PKG, load_golden

if PKG not in sys.path:
    sys.path.insert(0, PKG)

from phe import _native  # noqa: E402
from phe._engine import random_lt_n_limbs  # noqa: E402
from phe.codec import EncodedNumber as E  # noqa: E402


class Key:
    def __init__(self, n):
        self.n, self.max_int = n, n // 3 - 1


@pytest.fixture(scope="module")
def pk():
    return Key(int(load_golden(2048)["n"], 16))


def floats():
    rng = np.random.default_rng(5)
    x = rng.standard_normal(3000) * 10.0 ** rng.integers(-20, 20, 3000)
    return np.concatenate([x, [0.0, -0.0, 1.0, -1.0, 2.0 ** -1074, 1e300, -1e300, 0.1, -0.1, 2.0 ** 52 + 1, 1 / 3]])


def test_encode_arrays_equal_scalar_encode(pk):
    x = floats()
    mag, neg, exps = E.encode_signed(x)
    got = _native.limbs_to_ints(E.signed_to_limbs(pk, mag, neg, 64))
    for v, enc, e in zip(x.tolist(), got, exps.tolist()):
        ref = E.encode(pk, v)
        assert (ref.encoding, ref.exponent) == (enc, e), v
    for arr in (np.array([0, 1, -1, 2 ** 63 - 1, -2 ** 63, 123456789, -987654321], dtype=np.int64),
                np.array([0, 2 ** 64 - 1, 5], dtype=np.uint64), np.array([-128, 127, 0], dtype=np.int8)):
        mag, neg, exps = E.encode_signed(arr)
        assert not exps.any()
        got = _native.limbs_to_ints(E.signed_to_limbs(pk, mag, neg, 64))
        assert got == [E.encode(pk, int(v)).encoding for v in arr.tolist()]
    # homogeneous Python lists encode like the array of the same dtype; what the array form declines (mixed lists:
    # an int and the float of the same value get different exponents; precision / max_exponent) goes element by element
    for lst in ([1.0, -2.5, 1e-9, 0.0], [3, -4, 0, 2 ** 62]):
        mag, neg, exps = E.encode_signed(lst)
        got = _native.limbs_to_ints(E.signed_to_limbs(pk, mag, neg, 64))
        assert [(E.encode(pk, v).encoding, E.encode(pk, v).exponent) for v in lst] == list(zip(got, exps.tolist()))
    assert E.encode_signed([1, 2.0]) is None and E.encode_signed([2 ** 70, 1]) is None and E.encode_signed([True, False]) is None
    assert E.encode_signed([np.float64(1.0), 2.0]) is None and E.encode_signed(x, precision=1e-3) is None
    assert E.encode_signed(np.array([1.0, 2.0], dtype=np.float32)) is None and E.encode_signed(x, max_exponent=-3) is None
    with pytest.raises(ValueError):
        E.encode_signed(np.array([1.0, np.inf]))


def test_negative_values_with_a_borrow_out_of_the_low_limbs():
    k = Key((1 << 2047) | (1 << 200) | 1)          # n's low 64 bits are 1: n - mag borrows for every mag > 1
    mag = np.array([5, 1, 2 ** 64 - 1, 0], dtype=np.uint64)
    limbs = E.signed_to_limbs(k, mag, np.ones(4, dtype=bool), 64)
    assert _native.limbs_to_ints(limbs) == [k.n - 5, k.n - 1, k.n - (2 ** 64 - 1), 0]
    assert E.decode_limbs(k, limbs, [0, 0, 0, 0]) == [-5, -1, -(2 ** 64 - 1), 0]


def test_range_error_for_toy_keys():
    k = Key(1000003 * 999983)
    with pytest.raises(ValueError, match="Integer needs to be within"):
        E.signed_to_limbs(k, np.array([k.max_int + 1], dtype=np.uint64), np.zeros(1, dtype=bool), 4)


def test_decode_arrays_equal_scalar_decode(pk):
    x = floats()
    mag, neg, exps = E.encode_signed(x)
    limbs = E.signed_to_limbs(pk, mag, neg, 64)
    got = E.decode_limbs(pk, limbs, exps)
    ref = [E(pk, v, int(e)).decode() for v, e in zip(_native.limbs_to_ints(limbs), exps)]
    assert all(a == b and type(a) is type(b) for a, b in zip(got, ref))
    assert got == x.tolist()
    xi = np.array([0, 1, -1, 2 ** 63 - 1, -2 ** 63], dtype=np.int64)
    mag, neg, exps = E.encode_signed(xi)
    got = E.decode_limbs(pk, E.signed_to_limbs(pk, mag, neg, 64), exps)
    assert got == xi.tolist() and all(type(a) is int for a in got)
    # rows the numpy path declines: wide mantissas (products), positive / very negative exponents
    vals = [3 ** 200, pk.n - 3 ** 200, 7, pk.n - 7, pk.n - 1, 1, 2 ** 64, pk.n - 2 ** 64]
    exs = [-30, -30, 2, -80, -3, -64, -5, 0]
    got = E.decode_limbs(pk, _native.ints_to_limbs(vals, 64), exs)
    ref = [E(pk, v, e).decode() for v, e in zip(vals, exs)]
    assert got == ref and [type(a) for a in got] == [type(b) for b in ref]
    # errors are the scalar decoder's
    for bad, err in ((pk.n // 2, OverflowError), (pk.n, ValueError), (pk.n + 5, ValueError)):
        with pytest.raises(err):
            E.decode_limbs(pk, _native.ints_to_limbs([bad], 64), [0])


@pytest.mark.parametrize("n", [3, 5, 2 ** 32 + 1, 2 ** 64 - 59, (1 << 100) + 7, None])
def test_random_lt_n_limbs(n, pk):
    n = pk.n if n is None else n
    limbs = (n.bit_length() + 31) // 32 + (0 if n > 2 ** 200 else 1)
    draws = _native.limbs_to_ints(random_lt_n_limbs(n, 4000, limbs))
    assert all(0 < v < n for v in draws)
    if n < 10:
        assert set(draws) == set(range(1, n))
    else:
        assert len(set(draws)) > 3900 and max(draws) > n // 2     # spread over the range, top limb included
    buf = np.empty((7, limbs), dtype=np.uint32)
    assert random_lt_n_limbs(n, 7, limbs, out=buf) is buf
    assert random_lt_n_limbs(n, 0, limbs).shape == (0, limbs)
