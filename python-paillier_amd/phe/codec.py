"""Fixed-point codec between Python numbers and Paillier plaintexts (integers mod n).

Behavioural mirror of the reference's phe/encoding.py (EncodedNumber.encode :110-199, decode :201-233,
decrease_exponent_to :235-265) — same class name, attributes, results and exceptions — plus array forms
(`encode_many` / `decode_many`) that feed the batched GPU entry points.  The codec itself is host-side
integer bookkeeping; the heavy work it feeds is on the GPU.

Representation: value = mantissa * BASE**exponent with mantissa stored mod n; residues <= max_int are
non-negative, residues >= n - max_int are negative, the middle third flags overflow.
"""
import fractions
import math
import sys

import numpy as np


class EncodedNumber(object):
    BASE = 16
    LOG2_BASE = math.log(BASE, 2)
    FLOAT_MANTISSA_BITS = sys.float_info.mant_dig

    def __init__(self, public_key, encoding, exponent):
        self.public_key = public_key
        self.encoding = encoding
        self.exponent = exponent

    # ---- exponent selection (phe/encoding.py:160-183) ------------------------------------------
    @classmethod
    def _natural_exponent(cls, scalar, precision):
        if precision is not None:
            return math.floor(math.log(precision, cls.BASE))
        if isinstance(scalar, int):
            return 0
        if isinstance(scalar, float):
            lsb_power_of_two = math.frexp(scalar)[1] - cls.FLOAT_MANTISSA_BITS
            return math.floor(lsb_power_of_two / cls.LOG2_BASE)
        raise TypeError("Don't know the precision of type %s." % type(scalar))

    @classmethod
    def encode(cls, public_key, scalar, precision=None, max_exponent=None):
        exponent = cls._natural_exponent(scalar, precision)
        if max_exponent is not None:
            exponent = min(max_exponent, exponent)
        # exact rational arithmetic, then round-half-even like the reference (phe/encoding.py:191-192)
        scaled = fractions.Fraction(scalar) * fractions.Fraction(cls.BASE) ** -exponent
        int_rep = round(scaled)
        if abs(int_rep) > public_key.max_int:
            raise ValueError('Integer needs to be within +/- %d but got %d' % (public_key.max_int, int_rep))
        return cls(public_key, int_rep % public_key.n, exponent)

    def _signed_mantissa(self):
        pk = self.public_key
        if self.encoding >= pk.n:
            raise ValueError('Attempted to decode corrupted number')
        if self.encoding <= pk.max_int:
            return self.encoding
        if self.encoding >= pk.n - pk.max_int:
            return self.encoding - pk.n
        raise OverflowError('Overflow detected in decrypted number')

    def decode(self):
        mantissa = self._signed_mantissa()
        if self.exponent >= 0:
            return mantissa * self.BASE ** self.exponent
        try:
            return mantissa / self.BASE ** -self.exponent
        except OverflowError as e:
            raise OverflowError('decoded result too large for a float') from e

    def decrease_exponent_to(self, new_exp):
        if new_exp > self.exponent:
            raise ValueError('New exponent %i should be more negative thanold exponent %i' % (new_exp, self.exponent))
        factor = pow(self.BASE, self.exponent - new_exp)
        return self.__class__(self.public_key, self.encoding * factor % self.public_key.n, new_exp)

    # ---- array forms ------------------------------------------------------------------------------
    @classmethod
    def encode_many(cls, public_key, values, precision=None, max_exponent=None):
        """Encode a sequence / numpy array.  Returns (encodings: list[int], exponents: list[int]).

        dtype rules: float arrays follow the float rule per element (own exponent each),
        integer arrays get exponent 0.  float64 arrays with BASE a power of two and no precision /
        max_exponent take a vectorised path (exact: the mantissa shift is < log2(BASE) bits); anything
        else goes through `encode` element by element."""
        arr = values if isinstance(values, np.ndarray) else None
        n, max_int = public_key.n, public_key.max_int
        log2b = int(round(cls.LOG2_BASE))
        pow2_base = (1 << log2b) == cls.BASE
        if arr is not None and arr.dtype == np.float64 and precision is None and max_exponent is None and pow2_base:
            if not np.all(np.isfinite(arr)):
                raise ValueError("cannot encode inf / nan")
            mant, e2 = np.frexp(arr)
            lsb = e2.astype(np.int64) - cls.FLOAT_MANTISSA_BITS
            exps = np.floor_divide(lsb, log2b)
            shift = lsb - exps * log2b                       # 0 .. log2b-1
            imant = np.ldexp(mant, cls.FLOAT_MANTISSA_BITS).astype(np.int64)   # exact: |mant| < 1, 53 bits
            if log2b + cls.FLOAT_MANTISSA_BITS <= 62:
                reps = (imant << shift).tolist()
            else:
                reps = [int(a) << int(s) for a, s in zip(imant.tolist(), shift.tolist())]
            worst = max((abs(v) for v in reps), default=0)
            if worst > max_int:
                raise ValueError('Integer needs to be within +/- %d but got %d' % (max_int, worst))
            return [v % n for v in reps], exps.tolist()
        if arr is not None and arr.dtype.kind in "iu" and precision is None:
            reps = arr.tolist()
            exps = [0] * len(reps)
            if max_exponent is not None and max_exponent < 0:
                scale = cls.BASE ** -max_exponent
                reps = [v * scale for v in reps]
                exps = [max_exponent] * len(reps)
            worst = max((abs(v) for v in reps), default=0)
            if worst > max_int:
                raise ValueError('Integer needs to be within +/- %d but got %d' % (max_int, worst))
            return [v % n for v in reps], exps
        seq = arr.tolist() if arr is not None else list(values)
        encs, exps = [], []
        for v in seq:
            e = v if isinstance(v, EncodedNumber) else cls.encode(public_key, v, precision, max_exponent)
            encs.append(e.encoding)
            exps.append(e.exponent)
        return encs, exps

    # ---- limb-array forms: no Python integer per element ----------------------------------
    @classmethod
    def encode_signed(cls, values, precision=None, max_exponent=None):
        """float64 / integer numpy array -> (magnitude uint64, negative bool, exponents int64) with
        value = (-1)^negative * magnitude * BASE^exponent, element-wise the int_rep / exponent `encode` picks
        (phe/encoding.py:160-199).  None when the array form does not apply (other dtypes, precision / max_exponent
        given, BASE not a power of two): the caller then encodes element by element."""
        if precision is not None or max_exponent is not None:
            return None
        if isinstance(values, (list, tuple)) and len(values):
            # a homogeneous list of Python floats, or of Python ints that fit int64, encodes like the array of the same
            # dtype (mixed lists do not: an int and the float of the same value get different exponents)
            kind = type(values[0])
            if kind is float and all(type(v) is float for v in values):
                values = np.array(values, dtype=np.float64)
            elif kind is int and all(type(v) is int for v in values):
                try:
                    values = np.array(values, dtype=np.int64)
                except OverflowError:
                    return None
        if not isinstance(values, np.ndarray):
            return None
        log2b = int(round(cls.LOG2_BASE))
        if (1 << log2b) != cls.BASE or log2b + cls.FLOAT_MANTISSA_BITS > 62:
            return None
        arr = values.reshape(-1)
        if arr.dtype == np.float64:
            if not np.all(np.isfinite(arr)):
                raise ValueError("cannot encode inf / nan")
            mant, e2 = np.frexp(arr)
            lsb = e2.astype(np.int64) - cls.FLOAT_MANTISSA_BITS
            exps = np.floor_divide(lsb, log2b)
            shift = lsb - exps * log2b                                           # 0 .. log2b-1
            rep = np.ldexp(mant, cls.FLOAT_MANTISSA_BITS).astype(np.int64) << shift   # exact: |mant| < 1, 53 bits
            neg = rep < 0
            return np.abs(rep).astype(np.uint64), neg, exps
        if arr.dtype.kind == "u":
            return arr.astype(np.uint64), np.zeros(arr.shape, dtype=bool), np.zeros(arr.shape, dtype=np.int64)
        if arr.dtype.kind == "i":
            rep = arr.astype(np.int64)
            neg = rep < 0
            mag = rep.astype(np.uint64)
            mag[neg] = (~mag[neg]) + np.uint64(1)                                 # two's complement: safe for -2^63
            return mag, neg, np.zeros(arr.shape, dtype=np.int64)
        return None

    @staticmethod
    def signed_to_limbs(public_key, mag, neg, n_limbs):
        """(-1)^neg * mag mod n as (len, n_limbs) little-endian uint32 rows, magnitudes below 2^64 (n_limbs >= 3).
        Raises the reference's range error if a magnitude exceeds max_int (only possible for toy keys)."""
        mag = np.ascontiguousarray(mag, dtype=np.uint64)
        n = public_key.n
        if len(mag) and public_key.max_int < (1 << 64):
            worst = int(mag.max())
            if worst > public_key.max_int:
                raise ValueError('Integer needs to be within +/- %d but got %d' % (public_key.max_int, worst))
        neg = np.asarray(neg, dtype=bool) & (mag != 0)
        n_arr = np.frombuffer(n.to_bytes(4 * n_limbs, "little"), dtype=np.uint32)
        n_lo = np.uint64(n & 0xffffffffffffffff)
        out = np.empty((len(mag), n_limbs), dtype=np.uint32)
        # limbs 2..: n's own where the value is negative (n - mag with mag < 2^64), zero elsewhere — written in place
        np.multiply(neg.astype(np.uint32)[:, None], n_arr[None, 2:], out=out[:, 2:])
        lo = np.where(neg, n_lo - mag, mag)                    # uint64, wraps when the subtraction borrows
        out[:, 0] = (lo & np.uint64(0xffffffff)).astype(np.uint32)
        out[:, 1] = (lo >> np.uint64(32)).astype(np.uint32)
        for i in np.nonzero(neg & (mag > n_lo))[0].tolist():   # a borrow out of the low 64 bits: rare, exact by hand
            out[i] = np.frombuffer((n - int(mag[i])).to_bytes(4 * n_limbs, "little"), dtype=np.uint32)
        return out

    @staticmethod
    def signed_limbs_to_plain(public_key, mag_limbs, neg, n_limbs):
        """(-1)^neg * value mod n for magnitudes given as (len, w) little-endian uint32 rows (w < n_limbs, every
        value <= max_int: the caller has checked the bit lengths) -> (len, n_limbs) rows: value, or n - value."""
        mag_limbs = np.ascontiguousarray(mag_limbs, dtype=np.uint32)
        count, w = mag_limbs.shape
        out = np.zeros((count, n_limbs), dtype=np.uint32)
        out[:, :w] = mag_limbs
        rows = np.nonzero(np.asarray(neg, dtype=bool) & mag_limbs.any(axis=1))[0]
        if len(rows):
            n_arr = np.frombuffer(public_key.n.to_bytes(4 * n_limbs, "little"), dtype=np.uint32).astype(np.int64)
            res = np.broadcast_to(n_arr, (len(rows), n_limbs)).copy()
            res[:, :w] -= mag_limbs[rows].astype(np.int64)
            borrow = np.zeros(len(rows), dtype=np.int64)
            for j in range(n_limbs):                              # the borrow dies out right above the magnitude's limbs
                col = res[:, j] - borrow
                borrow = (col < 0).astype(np.int64)
                res[:, j] = col + (borrow << 32)
                if j >= w and not borrow.any():
                    break
            out[rows] = res.astype(np.uint32)
        return out

    @classmethod
    def decode_limbs(cls, public_key, limbs, exponents):
        """Plaintext rows (len, n_limbs) + exponents -> list of numbers, element-wise identical to
        cls(public_key, value, exponent).decode().  Rows whose magnitude fits 64 bits and whose exponent is in
        -64..0 are decoded with numpy — no Python integer per element —; everything else (and every error) goes
        through `decode`.

        This runs on the host while the GPU decrypts the next chunk (keys.py decrypt_batch): it has to stay under the
        kernel time of a chunk (~29 ms per 65,536 rows of a 2048-bit key), so the classification reads every row ONCE
        as 64-bit words (an OR-reduction; the comparison with n's high words only for the rows that are not small
        non-negative numbers) and the values leave numpy through one `tolist` per kind."""
        limbs = np.ascontiguousarray(limbs, dtype=np.uint32)
        exps = np.asarray(exponents, dtype=np.int64).reshape(-1)
        count, n_limbs = limbs.shape
        n = public_key.n
        if not (n_limbs >= 4 and n_limbs % 2 == 0 and n.bit_length() > 32 * (n_limbs - 1) and count):
            from . import _native
            return [cls(public_key, v, int(e)).decode() for v, e in zip(_native.limbs_to_ints(limbs), exps.tolist())]
        words = limbs.view(np.uint64)                          # (count, n_limbs / 2) little-endian 64-bit words
        lo = words[:, 0]
        high = np.bitwise_or.reduce(words[:, 1:], axis=1)
        pos = high == 0                                        # value < 2^64 <= max_int
        # negative with |value| < 2^64 and no borrow out of the low 64 bits: the high words are n's, the low one is below n's
        neg = np.zeros(count, dtype=bool)
        cand = np.nonzero(~pos)[0]
        n_lo = np.uint64(n & 0xffffffffffffffff)
        if len(cand):
            n_words = np.frombuffer(n.to_bytes(4 * n_limbs, "little"), dtype=np.uint64)
            sub = words if len(cand) == count else words[cand]
            same = np.bitwise_or.reduce(sub[:, 1:] ^ n_words[None, 1:], axis=1) == 0
            neg[cand] = same & (sub[:, 0] < n_lo)
        in_range = (exps <= 0) & (exps >= -64)
        fast = (pos | neg) & in_range
        mag = np.where(neg, n_lo - lo, lo)
        is_float = exps < 0
        # integers (exponent 0) beyond int64 take the slow path: numpy has no signed type for them
        fast &= is_float | (mag < np.uint64(1 << 63))
        out = None
        if fast.all() and (is_float.all() or not is_float.any()):
            # the common case — one kind, every row small: no index arrays, one tolist
            if is_float[0]:
                # correctly rounded uint64 -> float64, then an exact power-of-two scaling = the reference's true division
                vals = np.ldexp(mag.astype(np.float64), exps * int(round(cls.LOG2_BASE)))
                np.negative(vals, out=vals, where=neg)
            else:
                vals = mag.astype(np.int64)
                np.negative(vals, out=vals, where=neg)
            return vals.tolist()
        out = [None] * count
        fl = np.nonzero(fast & is_float)[0]
        if len(fl):
            vals = np.ldexp(mag[fl].astype(np.float64), exps[fl] * int(round(cls.LOG2_BASE)))
            np.negative(vals, out=vals, where=neg[fl])
            for i, v in zip(fl.tolist(), vals.tolist()):
                out[i] = v
        it = np.nonzero(fast & ~is_float)[0]
        if len(it):
            vals = mag[it].astype(np.int64)
            np.negative(vals, out=vals, where=neg[it])
            for i, v in zip(it.tolist(), vals.tolist()):
                out[i] = v
        slow = np.nonzero(~fast)[0]
        if len(slow):
            from . import _native
            for i, v in zip(slow.tolist(), _native.limbs_to_ints(limbs[slow])):
                out[i] = cls(public_key, v, int(exps[i])).decode()
        return out

    @classmethod
    def decode_many(cls, public_key, encodings, exponents):
        """Inverse of encode_many: list of ints/floats, element-wise identical to
        cls(public_key, enc, exp).decode()."""
        return [cls(public_key, enc, exp).decode() for enc, exp in zip(encodings, exponents)]
