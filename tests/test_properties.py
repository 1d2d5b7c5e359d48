"""Property tests (hypothesis) of the host-side array code and of the small device routines that run on the CPU build:
whatever the inputs, the array forms must equal the element-by-element definitions they replace."""
import sys

import numpy as np
from hypothesis import Phase, example, given, settings, strategies as st

from conftest import PKG, load_golden

if PKG not in sys.path:
    sys.path.insert(0, PKG)

from phe import _native  # noqa: E402
from phe._engine import Engine  # noqa: E402
from phe.codec import EncodedNumber as E  # noqa: E402

FAST = settings(max_examples=60, deadline=None)


class _Key:                                                   # what the codec needs of a public key
    def __init__(self, n):
        self.n = n
        self.max_int = n // 3 - 1


_KEY = _Key(int(load_golden(2048)["n"], 16))
_EMU = None


def _emu():
    global _EMU
    if _EMU is None:
        from emu_lib import Emu
        _EMU = Emu()
    return _EMU


@FAST
@given(st.lists(st.tuples(st.integers(0, 2 ** 64 - 1), st.integers(0, 200)), min_size=1, max_size=40))
@example([(0, 96), (1, 0)])                 # round-1 regression: a zero magnitude carrying the largest shift
@example([(0, 64), (1, 0)])
@example([(0, 200), (2 ** 64 - 1, 31), (0, 0)])
def test_shifted_limbs_is_a_left_shift(pairs):
    mags, shifts = zip(*pairs)
    limbs, bits = Engine.shifted_limbs(np.array(mags, dtype=np.uint64), np.array(shifts, dtype=np.int64))
    want = [m << s for m, s in zip(mags, shifts)]
    assert _native.limbs_to_ints(limbs) == want
    assert bits == max(1, max(w.bit_length() for w in want))


@FAST
@given(st.lists(st.floats(allow_nan=False, allow_infinity=False, width=64, min_value=-1e150, max_value=1e150), min_size=1, max_size=30))
def test_float_arrays_and_lists_encode_like_scalars(values):
    for form in (np.array(values, dtype=np.float64), list(values)):
        mag, neg, exps = E.encode_signed(form)
        got = _native.limbs_to_ints(E.signed_to_limbs(_KEY, mag, neg, 64))
        want = [E.encode(_KEY, float(v)) for v in values]
        assert got == [w.encoding for w in want] and exps.tolist() == [w.exponent for w in want]


@FAST
@given(st.lists(st.integers(-2 ** 63, 2 ** 63 - 1), min_size=1, max_size=30))
def test_int_lists_encode_like_scalars(values):
    mag, neg, exps = E.encode_signed(list(values))
    got = _native.limbs_to_ints(E.signed_to_limbs(_KEY, mag, neg, 64))
    assert got == [E.encode(_KEY, v).encoding for v in values] and not exps.any()


@FAST
@given(st.lists(st.tuples(st.integers(0, 2 ** 150), st.booleans()), min_size=1, max_size=30))
def test_signed_limbs_to_plain_is_value_or_n_minus_value(pairs):
    vals, neg = zip(*pairs)
    got = E.signed_limbs_to_plain(_KEY, _native.ints_to_limbs(list(vals), 5), np.array(neg), 64)
    assert _native.limbs_to_ints(got) == [(_KEY.n - v) if (s and v) else v for v, s in zip(vals, neg)]


@FAST
@given(st.integers(1, 12), st.data())
def test_decimal_conversion_is_str_and_int(words, data):
    xs = data.draw(st.lists(st.integers(0, 2 ** (32 * words) - 1), min_size=1, max_size=12))
    emu = _emu()
    digits = emu.to_decimal(_native.ints_to_limbs(xs, words))
    assert [bytes(r).decode().lstrip("0") or "0" for r in digits] == [str(x) for x in xs]
    assert _native.limbs_to_ints(emu.from_decimal(digits, words)) == xs


@settings(max_examples=25, deadline=None)
@given(st.lists(st.tuples(st.integers(2, 2 ** 63 - 1).map(lambda v: 2 * v + 1), st.integers(2, 2 ** 30)), min_size=1, max_size=12))
def test_miller_rabin_rows_equal_the_textbook_round(pairs):
    ns = [n for n, _ in pairs]
    bases = [2 + a % (n - 3) for n, a in pairs]

    def spp(n, a):
        d, s = n - 1, 0
        while d % 2 == 0:
            d, s = d // 2, s + 1
        x = pow(a, d, n)
        if x in (1, n - 1):
            return True
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                return True
        return False
    got = _emu().miller_rabin(_native.ints_to_limbs(ns, 2), _native.ints_to_limbs(bases, 2))
    assert got.tolist() == [spp(n, a) for n, a in zip(ns, bases)]


# magnitudes a 256-bit test key can align (an exponent gap of more than ~60 makes BASE^gap exceed n: ValueError in the
# scalar API and in the vector API alike)
_finite = st.floats(allow_nan=False, allow_infinity=False, width=64, min_value=-1e6, max_value=1e6).filter(
    lambda v: v == 0 or abs(v) > 1e-6)
_scalar = st.one_of(_finite, st.integers(-10 ** 9, 10 ** 9))


@settings(max_examples=12, deadline=None, phases=[Phase.explicit, Phase.reuse, Phase.generate])
@given(st.lists(st.tuples(_finite, _scalar, _scalar), min_size=1, max_size=5), st.booleans())
def test_vector_operators_equal_the_scalar_operators(rows, as_array):
    """EncryptedVector +, -, *, dot on the emulator backend against EncryptedNumber's operators (the drop-in scalar API,
    pinned to the reference by tests/test_api.py): same ciphertext bits, same exponents, element by element"""
    import emu_backend
    emu_backend.install()
    from phe import paillier
    g = load_golden(256)
    pub = paillier.PaillierPublicKey(int(g["n"], 16))
    xs = [r[0] for r in rows]
    ws = [r[1] for r in rows]
    vs = [r[2] for r in rows]
    r_values = list(range(3, 3 + len(rows)))
    vec = pub.encrypt_batch(np.array(xs), r_values=r_values)
    singles = [pub.encrypt(float(x), r_value=r) for x, r in zip(xs, r_values)]
    homogeneous = len({type(w) for w in ws}) == 1
    w_operand = np.array(ws) if (as_array and homogeneous) else ws
    v_operand = np.array(vs) if (as_array and len({type(v) for v in vs}) == 1) else vs

    def same(vector, numbers):
        assert vector.ciphertexts(False) == [x.ciphertext(False) for x in numbers]
        assert vector.exponents == [x.exponent for x in numbers]
    same(vec * w_operand, [s * w for s, w in zip(singles, ws)])
    same(vec + v_operand, [s + v for s, v in zip(singles, vs)])
    same(vec - v_operand, [s - v for s, v in zip(singles, vs)])
    prod = vec * w_operand
    same(vec + prod, [s + s * w for s, w in zip(singles, ws)])
    chain = None
    for s, w in zip(singles, ws):
        chain = s * w if chain is None else chain + s * w
    d = vec.dot(w_operand)
    assert (d.ciphertext(False), d.exponent) == (chain.ciphertext(False), chain.exponent)
