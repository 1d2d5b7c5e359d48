"""Ciphertext containers.

EncryptedNumber keeps the reference's scalar object API verbatim (phe/paillier.py:442-751: operators,
lazy obfuscation state machine, exponent alignment, error types, even the name-mangled private fields
the reference's tests poke at) but every bigint operation is a batch-of-one call into the GPU engine.

EncryptedVector is the batched sibling the GPU actually wants: one (batch, ct_limbs) uint32 limb array
plus per-element exponents and obfuscation flags; `+`, `*`, obfuscate(), decrease_exponent_to() are one
kernel launch each over the whole vector, with the same per-element semantics as EncryptedNumber.
"""
import numpy as np

from ._device import DeviceArray
from ._engine import random_lt_n, random_lt_n_limbs
from .codec import EncodedNumber


# matvec on a host vector gathers the columns a sparse matrix actually stores once the window tables of the whole
# vector would exceed this many bytes (tests lower it)
TABLE_COMPACT_BYTES = 64 << 20


def _check_alignment_factor(public_key, factor):
    """The reference aligns exponents with `self * pow(BASE, delta)` (phe/paillier.py:599): the factor goes through
    EncodedNumber.encode, which rejects integers above max_int (phe/encoding.py:194-196) — the vector forms raise the
    same error at the same bound instead of letting a factor in (max_int, n) through."""
    if factor > public_key.max_int:
        raise ValueError('Integer needs to be within +/- %d but got %d' % (public_key.max_int, factor))


class EncryptedNumber(object):
    def __init__(self, public_key, ciphertext, exponent=0):
        from .keys import PaillierPublicKey
        self.public_key = public_key
        self.__ciphertext = ciphertext
        self.exponent = exponent
        self.__is_obfuscated = False
        if isinstance(ciphertext, EncryptedNumber):
            raise TypeError('ciphertext should be an integer')
        if not isinstance(self.public_key, PaillierPublicKey):
            raise TypeError('public_key should be a PaillierPublicKey')

    # ---- operators ------------------------------------------------------------------------------
    def __add__(self, other):
        if isinstance(other, EncryptedNumber):
            return self._add_encrypted(other)
        if isinstance(other, EncodedNumber):
            return self._add_encoded(other)
        return self._add_scalar(other)

    def __radd__(self, other):
        return self.__add__(other)

    def __mul__(self, other):
        if isinstance(other, EncryptedNumber):
            raise NotImplementedError('Good luck with that...')
        encoding = other if isinstance(other, EncodedNumber) else EncodedNumber.encode(self.public_key, other)
        product = self._raw_mul(encoding.encoding)
        return EncryptedNumber(self.public_key, product, self.exponent + encoding.exponent)

    def __rmul__(self, other):
        return self.__mul__(other)

    def __sub__(self, other):
        return self + (other * -1)

    def __rsub__(self, other):
        return other + (self * -1)

    def __truediv__(self, scalar):
        return self.__mul__(1 / scalar)

    # ---- state ------------------------------------------------------------------------------------
    def ciphertext(self, be_secure=True):
        if be_secure and not self.__is_obfuscated:
            self.obfuscate()
        return self.__ciphertext

    def decrease_exponent_to(self, new_exp):
        if new_exp > self.exponent:
            raise ValueError('New exponent %i should be more negative than old exponent %i' % (new_exp, self.exponent))
        multiplied = self * pow(EncodedNumber.BASE, self.exponent - new_exp)
        multiplied.exponent = new_exp
        return multiplied

    def obfuscate(self):
        pk = self.public_key
        eng = pk._get_engine()
        pooled = None
        if hasattr(eng.ctx, "encrypt_dev"):
            from . import keys
            pooled = eng.take_obfuscators(1)
            if pooled is None and keys.SCALAR_POOL_REFILL:
                eng.fill_obfuscator_pool(keys.SCALAR_POOL_REFILL)          # see keys.SCALAR_POOL_REFILL
                pooled = eng.take_obfuscators(1)
        if pooled is not None:
            # an obfuscator r^n made ahead of time (PaillierPublicKey.precompute_obfuscators), used once: one product
            self.__ciphertext = eng.to_ints(eng.raw_add([self.__ciphertext % pk.nsquare], pooled.to_host()))[0]
        else:
            r = pk.get_random_lt_n()
            self.__ciphertext = eng.to_ints(eng.obfuscate([self.__ciphertext % pk.nsquare], [r]))[0]
        self.__is_obfuscated = True

    # ---- addition -----------------------------------------------------------------------------------
    def _add_scalar(self, scalar):
        encoded = EncodedNumber.encode(self.public_key, scalar, max_exponent=self.exponent)
        return self._add_encoded(encoded)

    def _add_encoded(self, encoded):
        if self.public_key != encoded.public_key:
            raise ValueError("Attempted to add numbers encoded against different public keys!")
        a, b = self, encoded
        if a.exponent > b.exponent:
            a = self.decrease_exponent_to(b.exponent)
        elif a.exponent < b.exponent:
            b = b.decrease_exponent_to(a.exponent)
        # E(b) with obfuscator 1 is 1 + n*b (phe/paillier.py:673): formed and multiplied in on the GPU
        pk = a.public_key
        eng = pk._get_engine()
        total = eng.to_ints(eng.add_plain([a.ciphertext(False)], [b.encoding]))[0]
        return EncryptedNumber(pk, total, a.exponent)

    def _add_encrypted(self, other):
        if self.public_key != other.public_key:
            raise ValueError("Attempted to add numbers encrypted against different public keys!")
        a, b = self, other
        if a.exponent > b.exponent:
            a = self.decrease_exponent_to(b.exponent)
        elif a.exponent < b.exponent:
            b = b.decrease_exponent_to(a.exponent)
        total = a._raw_add(a.ciphertext(False), b.ciphertext(False))
        return EncryptedNumber(a.public_key, total, a.exponent)

    def _raw_add(self, e_a, e_b):
        pk = self.public_key
        eng = pk._get_engine()
        return eng.to_ints(eng.raw_add([e_a % pk.nsquare], [e_b % pk.nsquare]))[0]

    def _raw_mul(self, plaintext):
        if not isinstance(plaintext, int):
            raise TypeError('Expected ciphertext to be int, not %s' % type(plaintext))
        pk = self.public_key
        if plaintext < 0 or plaintext >= pk.n:
            raise ValueError('Scalar out of bounds: %i' % plaintext)
        eng = pk._get_engine()
        return eng.to_ints(eng.raw_mul([self.ciphertext(False) % pk.nsquare], [plaintext]))[0]


def _fleet_min_rows():
    from . import fleet
    return fleet.MIN_ROWS_PER_DEVICE


class EncryptedVector(object):
    """A batch of Paillier ciphertexts under one public key, stored as little-endian uint32 limbs — either a
    host numpy array or a device-resident DeviceArray (`device=True` / `.to_device()`), in which case every
    operation below runs on HBM-resident operands and only plaintext-sized data crosses PCIe.

    A vector is a plain mutable container like the reference's EncryptedNumber: reading its rows may convert a lazy
    resident form back in place (`_limbs`), obfuscate() rewrites them — share one between threads only behind a lock of
    your own (the per-key engine serialises the native calls, not the vector's own fields)."""

    def __init__(self, public_key, limbs, exponents, obfuscated=False, _debt=0, _pair=False):
        self.public_key = public_key
        self.on_device = isinstance(limbs, DeviceArray)
        self._store = limbs if self.on_device else np.ascontiguousarray(limbs, dtype=np.uint32)
        # resident vectors only, two lazy forms of the rows; anything that looks at `_limbs` converts back first, so neither
        # ever leaves the vector:
        #   _debt: the rows hold x * R^-debt mod n^2 (Engine "Montgomery debt": a chain of additions on rows of 32-bit
        #          words costs one full-width Montgomery product per addition, the powers of R are settled at the end);
        #   _pair: the rows are in the engine's pair form (include/phe_hip.h "pair form": what the exponentiation kernels
        #          compute in) — an addition is ONE half-width pair product, nothing to settle, no word/limb conversion
        #          between two steps.  Entered by to_pair() and inherited by the results of `+` and sum().
        self._debt = int(_debt) if self.on_device else 0
        self._pair = bool(_pair) and self.on_device
        self._exps = np.array(exponents if isinstance(exponents, np.ndarray) else list(exponents), dtype=np.int64).reshape(-1)
        shape = self._store.shape
        if len(shape) != 2 or shape[0] != len(self._exps):
            raise ValueError("limbs must be (batch, ct_limbs) with one exponent per row")
        flags = obfuscated if isinstance(obfuscated, np.ndarray) else np.full(len(self._exps), bool(obfuscated))
        self._obfuscated = flags.astype(bool)

    def _eng(self):
        """the engine this vector's rows are worked on by: the key's ordinary engine, or — for a resident vector that lives on
        another device of a fleet (PaillierPublicKey.encrypt_batch_sharded, phe/fleet.py) — the engine of that device"""
        return self.public_key._engine_for(self._store) if self.on_device else self.public_key._get_engine()

    @property
    def _limbs(self):
        if self._pair:
            self._store = self._eng().from_pair_dev(self._store)
            self._pair = False
        elif self._debt:
            self._store = self._eng().scale_dev(self._store, self._debt)
            self._debt = 0
        return self._store

    @_limbs.setter
    def _limbs(self, value):
        self._store, self._debt, self._pair = value, 0, False

    def to_pair(self):
        """This vector, resident, with its rows in the engine's pair form (a no-op without the split-modulus engine): worth
        it for rows that will be added to several times — `+` between two resident vectors then costs one pair product, and
        sum() one call.  Everything else (download, decrypt, `*`, obfuscate, indexing) converts back by itself."""
        vec = self if self.on_device else self.to_device()
        eng = self._eng()
        if vec._pair or not eng.pair_form():
            return vec
        return EncryptedVector(self.public_key, eng.to_pair_dev(vec._limbs), vec._exps, vec._obfuscated.copy(), _pair=True)

    def _plain_rows(self):
        """the rows as plain ciphertext words WITHOUT touching the vector: the non-mutating readers (alignment inside `a + b`
        must not convert an operand out of the form its owner put it in, nor change `_store` under another thread's read)"""
        if self._pair:
            return self._eng().from_pair_dev(self._store)
        if self._debt:
            return self._eng().scale_dev(self._store, self._debt)
        return self._store

    def _pair_store(self):
        """the rows in pair form (converted if need be; the vector itself is left as it is)"""
        return self._store if self._pair else self._eng().to_pair_dev(self._plain_rows())

    @property
    def exponents(self):
        """per-row exponents as a list of ints (like EncryptedNumber.exponent); `exponent_array` is the numpy form"""
        return self._exps.tolist()

    @exponents.setter
    def exponents(self, value):
        value = np.array(list(value), dtype=np.int64).reshape(-1)
        if len(value) != len(self._exps):
            raise ValueError("one exponent per row")
        self._exps = value

    @property
    def exponent_array(self):
        return self._exps

    # ---- construction / movement -------------------------------------------------------------------------
    @classmethod
    def from_numbers(cls, public_key, numbers):
        numbers = list(numbers)
        for x in numbers:
            if not isinstance(x, EncryptedNumber):
                raise TypeError('Expected encrypted_number to be an EncryptedNumber not: %s' % type(x))
            if x.public_key != public_key:
                raise ValueError('encrypted_number was encrypted against a different key!')
        eng = public_key._get_engine()
        limbs = eng.cipher_limbs([x.ciphertext(False) % public_key.nsquare for x in numbers])
        flags = np.array([x._EncryptedNumber__is_obfuscated for x in numbers], dtype=bool)
        return cls(public_key, limbs, [x.exponent for x in numbers], flags)

    @classmethod
    def from_ciphertexts(cls, public_key, ciphertexts, exponents=0):
        ciphertexts = list(ciphertexts)
        if isinstance(exponents, int):
            exponents = [exponents] * len(ciphertexts)
        eng = public_key._get_engine()
        return cls(public_key, eng.cipher_limbs(ciphertexts), exponents)

    @classmethod
    def concatenate(cls, vectors):
        """One vector from several under the same key (e.g. the shards a batch was encrypted in): limbs, exponents and
        obfuscation flags in order.  Resident if every part is resident (device-to-device copies), else on the host."""
        vectors = list(vectors)
        if not vectors:
            raise ValueError("need at least one vector")
        pk = vectors[0].public_key
        if any(v.public_key != pk for v in vectors):
            raise ValueError("vectors were encrypted against different public keys")
        exps = np.concatenate([v._exps for v in vectors])
        flags = np.concatenate([v._obfuscated for v in vectors])
        if all(v.on_device for v in vectors):
            eng = pk._get_engine()
            out = DeviceArray(eng.ctx, len(exps), eng.ct_limbs)
            lo = 0
            for v in vectors:
                if len(v):
                    eng.ctx.d2d(out.ptr + lo * eng.ct_limbs * 4, v._limbs.ptr, v._limbs.nbytes)
                lo += len(v)
            eng.ctx.sync()
            return cls(pk, out, exps, flags)
        host = [v._limbs.to_host() if v.on_device else v._limbs for v in vectors]
        return cls(pk, np.concatenate(host), exps, flags)

    def to_device(self):
        if self.on_device:
            return self
        eng = self.public_key._get_engine()
        return EncryptedVector(self.public_key, eng.upload_cipher(self._limbs), self._exps, self._obfuscated.copy())

    def to_host(self):
        if not self.on_device:
            return self
        return EncryptedVector(self.public_key, self._limbs.to_host(), self._exps, self._obfuscated.copy())

    def _like(self, limbs, exponents, obfuscated=False):
        return EncryptedVector(self.public_key, limbs, exponents, obfuscated)

    def __len__(self):
        return len(self._exps)

    def __getitem__(self, i):
        if isinstance(i, (list, np.ndarray)):
            # index / boolean-mask selection like numpy's (a gather: the result lives on the host)
            idx = np.asarray(i)
            host = self._limbs.to_host() if self.on_device else self._limbs
            return self._like(np.ascontiguousarray(host[idx]), self._exps[idx], self._obfuscated[idx])
        if isinstance(i, slice):
            lo, hi, step = i.indices(len(self))
            if self.on_device and step == 1:
                return self._like(self._limbs.rows_view(lo, hi), self._exps[i], self._obfuscated[i])
            host = self._limbs.to_host() if self.on_device else self._limbs
            return self._like(host[i], self._exps[i], self._obfuscated[i])
        i = range(len(self))[i]
        row = self._limbs.rows_view(i, i + 1).to_host() if self.on_device else self._limbs[i:i + 1]
        eng = self._eng()
        x = EncryptedNumber(self.public_key, eng.to_ints(row)[0], int(self._exps[i]))
        x._EncryptedNumber__is_obfuscated = bool(self._obfuscated[i])
        return x

    def to_numbers(self):
        host = self.to_host()
        return [host[i] for i in range(len(self))]

    # ---- state -----------------------------------------------------------------------------------------
    def obfuscate(self, r_values=None):
        """c_i <- c_i * r_i^n mod n^2 for the rows not yet obfuscated (all rows if r_values is given)."""
        pk = self.public_key
        eng = self._eng()
        rows = np.arange(len(self)) if r_values is not None else np.nonzero(~self._obfuscated)[0]
        if len(rows) == 0:
            return self
        if r_values is None and hasattr(eng.ctx, "obfuscate_dev"):
            # fresh obfuscators as limb arrays straight from the CSPRNG (no Python integer per row)
            pooled = None
            if self.on_device and len(rows) == len(self):
                from . import keys
                take = eng.take_pool_rows if (self._pair and eng.pair_form()) else eng.take_obfuscators
                pooled = take(len(self))
                if pooled is None and len(self) <= keys.SCALAR_POOL_REFILL // 4:
                    eng.fill_obfuscator_pool(keys.SCALAR_POOL_REFILL)
                    pooled = take(len(self))
            if pooled is not None and self._pair and pooled.cols == self._store.cols:
                self._store = eng.pair_mul_dev(self._store, pooled)      # both in pair form: one pair product, stays there
            elif pooled is not None:
                self._limbs = eng.raw_add_dev(self._limbs, pooled)       # c * r^n with r^n made ahead of time, used once
            elif self.on_device:
                self._limbs = eng.obfuscate_fresh_dev(self._limbs, None if len(rows) == len(self) else rows)
            else:
                from ._engine import random_lt_n_limbs
                r = random_lt_n_limbs(pk.n, len(rows), eng.n_limbs)
                self._limbs[rows] = eng.obfuscate(np.ascontiguousarray(self._limbs[rows]), r)
            self._obfuscated[rows] = True
            return self
        r = list(r_values) if r_values is not None else random_lt_n(pk.n, len(rows))
        if self.on_device:
            # one launch over the whole vector; rows that need nothing get r = 1 (c * 1^n = c)
            full = [1] * len(self)
            for idx, rv in zip(rows.tolist(), r):
                full[idx] = rv
            self._limbs = eng.obfuscate_dev(self._limbs, full)
        else:
            self._limbs[rows] = eng.obfuscate(np.ascontiguousarray(self._limbs[rows]), r)
        self._obfuscated[rows] = True
        return self

    def limbs(self, be_secure=True):
        if be_secure:
            self.obfuscate()
        return self._limbs

    def ciphertexts(self, be_secure=True):
        limbs = self.limbs(be_secure)
        return self._eng().to_ints(limbs.to_host() if self.on_device else limbs)

    def decrease_exponent_to(self, new_exps):
        """Per-element EncryptedNumber.decrease_exponent_to: rows whose exponent is above the target are
        multiplied by BASE**delta (one variable-exponent modexp launch)."""
        old = self._exps
        new = np.broadcast_to(np.asarray(new_exps if isinstance(new_exps, (int, np.ndarray)) else list(new_exps),
                                         dtype=np.int64), old.shape).copy()
        up = np.nonzero(new > old)[0]
        if len(up):
            i = int(up[0])
            raise ValueError('New exponent %i should be more negative than old exponent %i' % (new[i], old[i]))
        rows = np.nonzero(new < old)[0]
        flags = self._obfuscated.copy()
        if len(rows) == 0:
            if self.on_device:                                # same rows, same lazy form: nothing to convert
                return EncryptedVector(self.public_key, self._store, new, flags, _debt=self._debt, _pair=self._pair)
            return self._like(self._limbs.copy(), new, flags)
        pk = self.public_key
        delta = (old - new)[rows]
        powers = {int(d): pow(EncodedNumber.BASE, int(d)) for d in np.unique(delta).tolist()}
        _check_alignment_factor(pk, max(powers.values()))
        flags[rows] = False
        eng = self._eng()
        log2b = int(round(EncodedNumber.LOG2_BASE))
        if (1 << log2b) == EncodedNumber.BASE and max(powers.values()) < pk.n - pk.max_int:
            # BASE^delta = 1 << (log2b * delta): the exponents as limb rows, no Python integer per row
            # (positive scalars below n - max_int: the plain powmod branch of _raw_mul, phe/paillier.py:751)
            if self.on_device:
                shifts = (old - new) * log2b                       # 0 for untouched rows: c^1 = c, still canonical
                e = eng.shifted_limbs(np.ones(len(old), dtype=np.uint64), shifts)
                return self._like(eng.powmod_dev(self._plain_rows(), e), new, flags)
            e, _ = eng.shifted_limbs(np.ones(len(rows), dtype=np.uint64), delta * log2b)
            limbs = self._limbs.copy()
            limbs[rows] = eng.ctx.powmod(np.ascontiguousarray(limbs[rows]), e)
            return self._like(limbs, new, flags)
        scal = [powers[d] for d in delta.tolist()]
        if self.on_device:
            # whole-vector launch: untouched rows are raised to the power 1 (c^1 = c, still canonical)
            exps = [1] * len(self)
            for i, sc in zip(rows.tolist(), scal):
                exps[i] = sc
            return self._like(eng.raw_mul_dev(self._plain_rows(), exps), new, flags)   # both branches of _raw_mul
        limbs = self._limbs.copy()
        limbs[rows] = eng.raw_mul(np.ascontiguousarray(limbs[rows]), scal)
        return self._like(limbs, new, flags)

    # ---- arithmetic --------------------------------------------------------------------------------------
    def _aligned(self, other_exps):
        target = np.minimum(self._exps, np.asarray(other_exps, dtype=np.int64))
        return self.decrease_exponent_to(target), target

    def _raw_add(self, a_limbs, b):
        eng = self._eng()
        if self.on_device:
            b_dev = b if isinstance(b, DeviceArray) else eng.upload_cipher(b)
            return eng.raw_add_dev(a_limbs, b_dev)
        b = b.to_host() if isinstance(b, DeviceArray) else b
        fl = self.public_key._get_fleet()
        if fl is not None and len(fl.shards(len(a_limbs), 8 * _fleet_min_rows())) > 1:
            # host rows over the devices of a fleet (phe/fleet.py): contiguous shards, one array back
            return np.concatenate(fl.run(len(a_limbs), lambda e, lo, hi: e.raw_add(a_limbs[lo:hi], b[lo:hi]), 8 * _fleet_min_rows()))
        return eng.raw_add(a_limbs, b)

    def __add__(self, other):
        pk = self.public_key
        if isinstance(other, EncryptedVector):
            if pk != other.public_key:
                raise ValueError("Attempted to add numbers encrypted against different public keys!")
            if len(other) != len(self):
                raise ValueError("vector lengths differ")
            if other.on_device != self.on_device:
                other = other.to_device() if self.on_device else other.to_host()
            if self.on_device and other.on_device and other._store.ctx is not self._store.ctx:
                # two resident vectors of a fleet that live on different devices: the other one comes over (through the host)
                host = other.to_host()
                other = EncryptedVector(pk, self._eng().upload_cipher(host._limbs), host._exps, host._obfuscated.copy())
            a, target = self._aligned(other._exps)
            b = other.decrease_exponent_to(target)
            eng = self._eng()
            if self.on_device and (a._pair or b._pair) and eng.pair_form():
                # one pair product (the operand that is not in pair form yet is converted: it pays once, the sum stays)
                return EncryptedVector(pk, eng.pair_mul_dev(a._pair_store(), b._pair_store()), target, _pair=True)
            if self.on_device and eng.lazy_products():
                # one Montgomery product; the missing powers of R are settled when the residues are looked at
                return EncryptedVector(pk, eng.montmul_dev(a._store, b._store), target, _debt=a._debt + b._debt + 1)
            return self._like(self._raw_add(a._limbs, b._limbs), target)
        # plain operand(s): scalar broadcast or sequence; encode with max_exponent = own exponent per row
        values = other if isinstance(other, (list, tuple, np.ndarray)) else [other] * len(self)
        if len(values) != len(self):
            raise ValueError("vector lengths differ")
        eng = self._eng()
        signed = EncodedNumber.encode_signed(values) if eng.n_limbs >= 4 else None
        if signed is not None:
            # array form of the loop below: encode(v, max_exponent=e) = the natural encoding with its mantissa
            # shifted down to min(natural exponent, e) — exact while the shifted magnitude still fits 64 bits
            mag, neg, nat = signed
            target = np.minimum(nat, self._exps)
            shift = (nat - target) * int(round(EncodedNumber.LOG2_BASE))
            bits = np.zeros(len(mag), dtype=np.int64)
            nz = mag != 0
            bits[nz] = np.floor(np.log2(mag[nz].astype(np.float64))).astype(np.int64) + 2   # upper bound on the bit length
            if np.all(bits + shift <= 63):
                a = self.decrease_exponent_to(target)
                plain = EncodedNumber.signed_to_limbs(pk, mag << shift.astype(np.uint64), neg, eng.n_limbs)
                limbs = eng.add_plain_dev(a._limbs, plain) if self.on_device else eng.add_plain(a._limbs, plain)
                return self._like(limbs, target)
            if int((bits + shift).max()) < min(pk.max_int.bit_length(), 32 * (eng.n_limbs - 1)):
                # wider than 64 bits after the shift (the plaintext's exponent is well above the ciphertext's):
                # the shifted mantissas as limb rows, n - value for the negative ones — still no Python integer per row
                a = self.decrease_exponent_to(target)
                wide, _ = eng.shifted_limbs(mag, shift)
                plain = EncodedNumber.signed_limbs_to_plain(pk, wide, neg, eng.n_limbs)
                limbs = eng.add_plain_dev(a._limbs, plain) if self.on_device else eng.add_plain(a._limbs, plain)
                return self._like(limbs, target)
        values = values.tolist() if isinstance(values, np.ndarray) else values
        encs, exps = [], []
        for v, e in zip(values, self.exponents):
            enc = v if isinstance(v, EncodedNumber) else EncodedNumber.encode(pk, v, max_exponent=e)
            if enc.public_key != pk:
                raise ValueError("Attempted to add numbers encoded against different public keys!")
            encs.append(enc)
            exps.append(enc.exponent)
        a, target = self._aligned(exps)
        plain = [enc.decrease_exponent_to(t).encoding for enc, t in zip(encs, target.tolist())]
        limbs = eng.add_plain_dev(a._limbs, plain) if self.on_device else eng.add_plain(a._limbs, plain)
        return self._like(limbs, target)

    __radd__ = __add__

    def __mul__(self, other):
        if isinstance(other, (EncryptedVector, EncryptedNumber)):
            raise NotImplementedError('Good luck with that...')
        pk = self.public_key
        values = other if isinstance(other, (list, tuple, np.ndarray)) else [other] * len(self)
        if len(values) != len(self):
            raise ValueError("vector lengths differ")
        eng = self._eng()
        signed = EncodedNumber.encode_signed(values) if eng.n_limbs >= 4 else None
        if signed is not None:
            mag, neg, exps = signed
            if self.on_device and self._pair and eng.pair_form() and not neg.any() and hasattr(eng, "pair_mul_scalars_dev"):
                # rows in the engine's pair form and no negative scalar (those take invert(c), phe/paillier.py:745-749, which
                # wants residues): the powers stay in the pair form — no conversion in, no exit
                return EncryptedVector(pk, eng.pair_mul_scalars_dev(self._store, mag), self._exps + exps, _pair=True)
            fl = None if self.on_device else self.public_key._get_fleet()
            if fl is not None and len(fl.shards(len(self))) > 1:
                rows = self._limbs
                limbs = np.concatenate(fl.run(len(self), lambda e, lo, hi: e.raw_mul_signed(rows[lo:hi], mag[lo:hi], neg[lo:hi])))
            else:
                limbs = eng.raw_mul_signed_dev(self._limbs, mag, neg) if self.on_device else eng.raw_mul_signed(self._limbs, mag, neg)
            return self._like(limbs, self._exps + exps)
        if isinstance(values, np.ndarray) or not any(isinstance(v, EncodedNumber) for v in values):
            encs, exps = EncodedNumber.encode_many(pk, values)
        else:
            pairs = [v if isinstance(v, EncodedNumber) else EncodedNumber.encode(pk, v) for v in values]
            encs, exps = [e.encoding for e in pairs], [e.exponent for e in pairs]
        limbs = eng.raw_mul_dev(self._limbs, encs) if self.on_device else eng.raw_mul(self._limbs, encs)
        return self._like(limbs, self._exps + np.asarray(exps, dtype=np.int64))

    __rmul__ = __mul__

    def __neg__(self):
        return self * -1

    def __sub__(self, other):
        if isinstance(other, EncryptedVector):
            return self + (other * -1)
        if isinstance(other, np.ndarray) and other.dtype.kind in "fi":
            return self + (-other)                                # stays on the array path of __add__
        if isinstance(other, (list, tuple, np.ndarray)):
            return self + [-v for v in (other.tolist() if isinstance(other, np.ndarray) else other)]
        return self + (-other)

    def __truediv__(self, scalar):
        return self * (1 / scalar)

    def sum(self):
        """Homomorphic sum of all elements -> EncryptedNumber (log2(batch) pairwise-product launches; on a
        device-resident vector the halves are views, nothing is copied or moved across PCIe)."""
        if len(self) == 0:
            raise ValueError("empty vector")
        pk = self.public_key
        eng = self._eng()
        cur = self.decrease_exponent_to(int(self._exps.min()))
        if self.on_device and cur._pair and eng.pair_form():
            # rows already in pair form: the whole pairwise tree is one call (log2 launches queued back to back), one exit
            root = eng.from_pair_dev(eng.pair_reduce_dev(cur._store))
            return EncryptedNumber(pk, eng.to_ints(root.to_host())[0], int(cur._exps[0]))
        if self.on_device and eng.lazy_products():
            # the pairwise tree at ONE Montgomery product per node: a level turns rows of debt d into rows of debt 2d + 1
            # (an odd row out is taken to the same debt by a product with a constant); settled once, at the root
            # every level is queued on the engine's launch stream, nothing is waited for until the root is there: a level's
            # output block holds one spare row, so that an unpaired last row joins it by ONE product written in place
            exp = int(cur._exps[0])
            root = eng.montmul_tree_dev(cur._store, cur._debt)   # (one serialised engine call: the tree and its settling product)
            return EncryptedNumber(pk, eng.to_ints(root.to_host())[0], exp)
        limbs, exp = cur._limbs, int(cur._exps[0])
        if self.on_device:
            while limbs.rows > 1:
                half = limbs.rows // 2
                merged = eng.raw_add_dev(limbs.rows_view(0, half), limbs.rows_view(half, 2 * half))
                if limbs.rows % 2:
                    grown = DeviceArray(eng.ctx, half + 1, limbs.cols)
                    eng.ctx.d2d(grown.ptr, merged.ptr, merged.nbytes)
                    eng.ctx.d2d(grown.ptr + merged.nbytes, limbs.rows_view(2 * half, 2 * half + 1).ptr, limbs.cols * 4)
                    eng.ctx.sync()
                    merged = grown
                limbs = merged
            return EncryptedNumber(pk, eng.to_ints(limbs.to_host())[0], exp)
        while limbs.shape[0] > 1:
            half = limbs.shape[0] // 2
            merged = eng.raw_add(np.ascontiguousarray(limbs[:half]), np.ascontiguousarray(limbs[half:2 * half]))
            limbs = np.concatenate([merged, limbs[2 * half:]]) if limbs.shape[0] % 2 else merged
        return EncryptedNumber(pk, eng.to_ints(limbs)[0], exp)

    # ---- bulk wire format (docs/serialisation.rst:24-43 of the reference) ----------------------------------
    def to_json(self, be_secure=True):
        """{"public_key": {"n": ...}, "values": [[str(ciphertext), exponent], ...]} — the reference's documented
        vector format; like EncryptedNumber.ciphertext() the rows are obfuscated first unless be_secure=False
        (one launch over the vector instead of one modexp per element).  The decimal strings come from the device
        (Engine.decimal_strings: the batch form of str(int)); the text is what json.dumps would write."""
        eng = self._eng()
        strings = eng.decimal_strings(self.limbs(be_secure))
        body = ", ".join('["%s", %d]' % (c, e) for c, e in zip(strings, self._exps.tolist()))
        return '{"public_key": {"n": %d}, "values": [%s]}' % (self.public_key.n, body)

    @classmethod
    def from_json(cls, text, device=False):
        import json
        from .keys import PaillierPublicKey
        doc = json.loads(text)
        public_key = PaillierPublicKey(n=int(doc["public_key"]["n"]))
        eng = public_key._get_engine()
        values = doc["values"]
        texts = [v[0] if isinstance(v[0], str) else str(int(v[0])) for v in values]
        limbs = eng.limbs_from_decimal(texts)                 # the batch form of int(str) per element
        vec = cls(public_key, limbs, [int(v[1]) for v in values])
        return vec.to_device() if device else vec

    def dot(self, plain):
        """sum_i self[i] * plain[i] -> EncryptedNumber: np.dot over ciphertexts (phe/tests/math_test.py:44-58,
        examples/logistic_regression_encrypted_model.py:170-177).  The ciphertext is bit-for-bit the one the chain of
        `*` (phe/paillier.py:721-751) and `+` (:705-719, with :570-601 decrease_exponent_to aligning the terms to
        the smallest exponent) produces — prod_i b_i^(e_i * BASE^delta_i) mod n^2, b_i = c_i or its inverse on the
        negative branch — but formed as ONE multi-exponentiation (Engine.raw_dot) instead of a modexp per element,
        a second one per misaligned element and log2(batch) product launches."""
        pk = self.public_key
        values = plain if isinstance(plain, (list, tuple, np.ndarray)) else [plain] * len(self)
        if len(values) != len(self):
            raise ValueError("vector lengths differ")
        if len(self) == 0:
            raise ValueError("empty vector")
        if any(isinstance(v, (EncryptedNumber, EncryptedVector)) for v in (values if not isinstance(values, np.ndarray) else ())):
            raise NotImplementedError('Good luck with that...')
        eng = self._eng()
        signed = EncodedNumber.encode_signed(values) if eng.n_limbs >= 4 else None
        if signed is not None:
            mag, neg, kexp = signed
        else:
            if isinstance(values, np.ndarray) or not any(isinstance(v, EncodedNumber) for v in values):
                encs, kexp = EncodedNumber.encode_many(pk, values)
            else:
                pairs = [v if isinstance(v, EncodedNumber) else EncodedNumber.encode(pk, v) for v in values]
                if any(e.public_key != pk for e in pairs):
                    raise ValueError("Attempted to multiply with numbers encoded against different public keys!")
                encs, kexp = [e.encoding for e in pairs], [e.exponent for e in pairs]
            threshold = pk.n - pk.max_int                     # phe/paillier.py:745: these take the inverted base
            for e in encs:
                if e < 0 or e >= pk.n:
                    raise ValueError('Scalar out of bounds: %i' % e)
            neg = np.fromiter((e >= threshold for e in encs), dtype=bool, count=len(encs))
            mag = [pk.n - e if e >= threshold else e for e in encs]
            kexp = np.asarray(kexp, dtype=np.int64)
        total = self._exps + kexp
        target = int(total.min())
        delta = total - target                                 # rows above the common exponent: * BASE^delta (:599)
        dmax = int(delta.max())
        if dmax:
            _check_alignment_factor(pk, pow(EncodedNumber.BASE, dmax))
        log2b = int(round(EncodedNumber.LOG2_BASE))
        if dmax == 0:
            exps = mag
        elif isinstance(mag, np.ndarray) and (1 << log2b) == EncodedNumber.BASE:
            exps = eng.shifted_limbs(mag, delta * log2b)         # k_i * BASE^delta_i as limb rows, no Python ints
        else:
            mags = mag.tolist() if isinstance(mag, np.ndarray) else mag
            powers = {d: pow(EncodedNumber.BASE, d) for d in np.unique(delta).tolist()}
            exps = [m * powers[d] for m, d in zip(mags, delta.tolist())]
        # rows resident in the pair form and no scalar on the negative branch: the multi-exponentiation takes them as they are
        in_form = self.on_device and self._pair and eng.pair_form() and hasattr(eng.ctx, "pair_multiexp_rows_dev") and not np.asarray(neg).any()
        return EncryptedNumber(pk, eng.raw_dot(self._store if in_form else self._limbs, exps, neg), target)

    # numpy must not try to broadcast over the rows of a vector: `W @ vec`, `w @ vec`, `vec @ w` come here
    __array_ufunc__ = None

    def __matmul__(self, other):
        """vec @ w (1-D plaintext) -> the dot product; vec @ M for a (len, cols) matrix -> M.T @ vec"""
        other = np.asarray(other) if not isinstance(other, np.ndarray) else other
        if other.ndim == 1:
            return self.dot(other)
        if other.ndim == 2:
            return self.matvec(np.ascontiguousarray(other.T))
        raise ValueError("matmul operand must be 1-D or 2-D")

    def __rmatmul__(self, other):
        """w @ vec -> the dot product; W @ vec for a (rows, len) matrix -> matvec(W)"""
        other = np.asarray(other) if not isinstance(other, np.ndarray) else other
        if other.ndim == 1:
            return self.dot(other)
        if other.ndim == 2:
            return self.matvec(other)
        raise ValueError("matmul operand must be 1-D or 2-D")

    def mean(self):
        """np.mean over ciphertexts (phe/tests/math_test.py:50-58): sum() times the encoded 1/len"""
        return self.sum() / len(self)

    def matvec(self, matrix):
        """matrix @ self for a plaintext (rows, len(self)) matrix — numpy array or scipy.sparse matrix — ->
        EncryptedVector of `rows` dot products, each bit for bit what `self.dot(matrix[r])` (hence the reference's chain
        of `*` and `+`) gives; for a sparse matrix the chain over the STORED entries of the row, which is what
        Bob.encrypted_score walks (examples/logistic_regression_encrypted_model.py:170-177: `_, idx = x.nonzero()`).
        Sparse matrices and dense ones with many rows go through the table-lookup multi-exponentiation
        (Engine.raw_matvec_csr: the window tables of every ciphertext once, then one ladder per row over its entries);
        few dense rows over a long vector through the chunked form (Engine.raw_matvec).  Negative-branch inverses are
        formed once for the whole vector.  A row without entries is the ciphertext 1 (an encryption of 0)."""
        pk = self.public_key
        eng = self._eng()
        if len(self) == 0:
            raise ValueError("empty vector")
        log2b = int(round(EncodedNumber.LOG2_BASE))
        pow2_base = (1 << log2b) == EncodedNumber.BASE
        sparse = hasattr(matrix, "tocsr") and not isinstance(matrix, np.ndarray)
        if sparse and eng.n_limbs >= 4 and pow2_base and eng.has_split_engine():
            csr = matrix.tocsr()
            if csr.shape[1] != len(self):
                raise ValueError("matrix must have shape (rows, %d)" % len(self))
            rows = csr.shape[0]
            data = np.ascontiguousarray(csr.data)
            signed = EncodedNumber.encode_signed(data) if len(data) else None
            indptr = np.asarray(csr.indptr, dtype=np.int64)
            cols = np.asarray(csr.indices, dtype=np.int64)
            if signed is not None or len(data) == 0:
                base_exp = int(self._exps.min())
                if len(data) == 0:
                    mag = np.zeros(0, np.uint64); neg = np.zeros(0, bool); total = np.zeros(0, np.int64)
                else:
                    mag, neg, kexp = signed
                    total = self._exps[cols] + kexp
                counts = np.diff(indptr)
                target = np.full(rows, base_exp, dtype=np.int64)
                filled = np.nonzero(counts)[0]
                if len(filled):
                    target[filled] = np.minimum.reduceat(total, indptr[filled])
                delta = total - np.repeat(target, counts)
                dmax = int(delta.max()) if len(delta) else 0
                if dmax:
                    _check_alignment_factor(pk, pow(EncodedNumber.BASE, dmax))
                exps = eng.shifted_limbs(mag, delta * log2b)
                base = self._limbs
                # the tables cover every ciphertext handed over: keep them to the columns that are actually stored
                # (host vectors: a gather; resident vectors are used whole, up to 8 GiB of tables)
                per_col = 2 * 15 * 2 * 4 * (eng.ct_limbs + 32)
                if not self.on_device and len(cols) and len(self) * per_col > TABLE_COMPACT_BYTES:
                    used, remap = np.unique(cols, return_inverse=True)
                    if len(used) < len(self):
                        base, cols = np.ascontiguousarray(base[used]), remap
                if base.shape[0] * per_col > (8 << 30):
                    raise MemoryError("matvec: window tables for %d ciphertexts would not fit the table budget; "
                                      "slice the matrix by columns" % base.shape[0])
                limbs = eng.raw_matvec_csr(base, indptr, cols, exps, neg, rows)
                return EncryptedVector(pk, limbs, target)
        if sparse:
            matrix = matrix.toarray()
        W = np.asarray(matrix)
        if W.ndim != 2 or W.shape[1] != len(self):
            raise ValueError("matrix must have shape (rows, %d)" % len(self))
        rows = W.shape[0]
        signed = EncodedNumber.encode_signed(np.ascontiguousarray(W)) if (eng.n_limbs >= 4 and rows) else None
        if signed is None:                                   # object / exotic dtypes: row by row through dot()
            return EncryptedVector.from_numbers(pk, [self.dot(list(row)) for row in W.tolist()]) if rows else self[:0]
        mag, neg, kexp = (a.reshape(W.shape) for a in signed)
        total = self._exps[None, :] + kexp
        target = total.min(axis=1)
        delta = total - target[:, None]
        dmax = int(delta.max())
        if dmax:
            _check_alignment_factor(pk, pow(EncodedNumber.BASE, dmax))
        # many rows over a vector whose tables fit comfortably (<= 1 GiB): build them once for all rows
        table_bytes = len(self) * 2 * 15 * 2 * 4 * (eng.ct_limbs + 32)
        if pow2_base and rows >= 64 and table_bytes <= (1 << 30) and eng.has_split_engine():
            exps = eng.shifted_limbs(mag.reshape(-1), (delta * log2b).reshape(-1))
            limbs = eng.raw_matvec_csr(self._limbs, None, None, exps, neg.reshape(-1), rows)
            return EncryptedVector(pk, limbs, target)
        if dmax == 0:
            exps = mag
        else:                                                    # k * BASE^delta as limb rows, no Python ints
            exps = eng.shifted_limbs(mag.reshape(-1), (delta * log2b).reshape(-1))
        in_form = self.on_device and self._pair and eng.pair_form() and hasattr(eng.ctx, "pair_multiexp_rows_dev") and not np.asarray(neg).any()
        limbs = eng.raw_matvec(self._store if in_form else self._limbs, exps, neg)
        return EncryptedVector(pk, limbs, target)
