"""Per-key engines: the only place where Python integers meet the C-ABI.

An Engine owns one native Context (include/phe_hip.h) for one key on one GPU and exposes the five
hot functions of the reference on *batches* of Python ints or uint32 limb arrays:

    reference (scalar, phe/paillier.py)                  here (batched, HIP)
    PaillierPublicKey.raw_encrypt      :102-139    ->    Engine.raw_encrypt
    EncryptedNumber.obfuscate          :603-624    ->    Engine.obfuscate
    PaillierPrivateKey.raw_decrypt     :328-374    ->    Engine.raw_decrypt
    EncryptedNumber._raw_add           :705-719    ->    Engine.raw_add
    EncryptedNumber._raw_mul           :721-751    ->    Engine.raw_mul   (both branches)

There is no CPU implementation behind these: if lib/libphe_hip.so or a GPU is missing, they raise.
"""
import os
import secrets

import numpy as np

from . import _native
from ._device import DeviceArray


def default_device():
    """One process per GPU: LOCAL_RANK picks the device unless PHE_HIP_DEVICE overrides it."""
    for var in ("PHE_HIP_DEVICE", "LOCAL_RANK"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            return int(v)
    return 0


def random_lt_n(n, count):
    """`count` cryptographically random integers in [1, n) as Python ints (see random_lt_n_limbs)."""
    limbs = (n.bit_length() + 31) // 32
    return _native.limbs_to_ints(random_lt_n_limbs(n, count, limbs)) if count else []


def random_lt_n_limbs(n, count, limbs):
    """`count` cryptographically random integers in [1, n) as a (count, limbs) uint32 array — the bulk form of
    PaillierPublicKey.get_random_lt_n (phe/paillier.py:141-143, random.SystemRandom().randrange(1, n)): same
    source (os.urandom), same distribution (rejection sampling on bit_length(n-1) bits), vectorised with numpy."""
    k = (n - 1).bit_length()
    top_limb, top_bits = (k - 1) // 32, k - 32 * ((k - 1) // 32)
    top_mask = np.uint32((1 << top_bits) - 1) if top_bits < 32 else np.uint32(0xffffffff)
    n_arr = _native.int_to_limbs(n, limbs)
    out = np.frombuffer(bytearray(secrets.token_bytes(count * limbs * 4)), dtype=np.uint32).reshape(count, limbs)
    rows = np.arange(count)
    while len(rows):
        out[rows, top_limb] &= top_mask
        if top_limb + 1 < limbs:
            out[rows, top_limb + 1:] = 0
        # rows that are >= n (lexicographic compare from the most significant limb down) or zero are redrawn
        lt = np.zeros(len(rows), dtype=bool)
        undecided = np.ones(len(rows), dtype=bool)
        for j in range(top_limb, -1, -1):
            col = out[rows, j]
            lt |= undecided & (col < n_arr[j])
            undecided &= (col == n_arr[j])
            if not undecided.any():
                break
        bad = ~lt
        bad[lt] = ~out[rows[lt]].any(axis=1) if len(rows) < count else ~out.any(axis=1)[lt]
        rows = rows[bad]
        if len(rows):
            out[rows] = np.frombuffer(secrets.token_bytes(len(rows) * limbs * 4), dtype=np.uint32).reshape(len(rows), limbs)
    return out


class Engine:
    def __init__(self, n, p=None, q=None, hp=None, hq=None, p_inverse=None, device=None):
        self.n = n
        self.nsquare = n * n
        self.max_int = n // 3 - 1
        self.device = default_device() if device is None else device
        self.ctx = _native.Context(n, p, q, hp, hq, p_inverse, device=self.device)
        self.n_limbs = self.ctx.n_limbs
        self.ct_limbs = self.ctx.ct_limbs

    # ---- limb helpers -------------------------------------------------------------------------
    def plain_limbs(self, ints):
        return _native.ints_to_limbs(ints, self.n_limbs)

    def cipher_limbs(self, ints):
        return _native.ints_to_limbs(ints, self.ct_limbs)

    @staticmethod
    def to_ints(arr):
        return _native.limbs_to_ints(arr)

    def _as_plain(self, x):
        return x if isinstance(x, np.ndarray) else self.plain_limbs(x)

    def _as_cipher(self, x):
        # Python ints are reduced mod n^2 on the way in (the reference's mulmod/powmod accept any int and
        # reduce implicitly; the kernels want residues)
        return x if isinstance(x, np.ndarray) else self.cipher_limbs([v % self.nsquare for v in x])

    # ---- the five hot functions (limb arrays in, limb arrays out) ------------------------------
    def raw_encrypt(self, m, r):
        """(1 + n*m) * r^n mod n^2 per row.  m is reduced mod n on the way in (value-identical to the
        reference's `% nsquare`, which lets m = n, n+1 wrap to 0, 1: phe/tests/paillier_test.py:114-126)."""
        if not isinstance(m, np.ndarray):
            m = self.plain_limbs([v % self.n for v in m])
        return self.ctx.encrypt(m, self._as_plain(r))

    def obfuscate(self, c, r):
        return self.ctx.obfuscate(self._as_cipher(c), self._as_plain(r))

    def raw_decrypt(self, c):
        return self.ctx.decrypt(self._as_cipher(c))

    def raw_add(self, a, b):
        return self.ctx.mulmod(self._as_cipher(a), self._as_cipher(b))

    def add_plain(self, c, plaintexts):
        """c * (1 + n*m) mod n^2 per row: E(a) + b for a plaintext encoding b (phe/paillier.py:673-675)."""
        if not isinstance(plaintexts, np.ndarray):
            plaintexts = self.plain_limbs([v % self.n for v in plaintexts])
        return self.ctx.add_plain(self._as_cipher(c), plaintexts)

    def raw_mul(self, c, scalars):
        """Per row: powmod(c, s, n^2) if s < n - max_int, else powmod(invert(c, n^2), n - s, n^2) — the same
        partition as phe/paillier.py:745-751 (the two branches give different ciphertext bits).
        `scalars` are Python ints already validated to lie in [0, n)."""
        c = self._as_cipher(c)
        n, threshold = self.n, self.n - self.max_int
        neg = [i for i, s in enumerate(scalars) if s >= threshold]
        exps = [n - s if s >= threshold else s for s in scalars]
        base = c
        if neg:
            base = c.copy()
            idx = np.asarray(neg, dtype=np.int64)
            base[idx] = self.ctx.invert(np.ascontiguousarray(c[idx]))
        width = max(1, (max(e.bit_length() for e in exps) + 31) // 32) if exps else 1
        return self.ctx.powmod(base, _native.ints_to_limbs(exps, width))

    def powmod_n2(self, base, exps):
        exps = list(exps)
        width = max(1, (max(e.bit_length() for e in exps) + 31) // 32) if exps else 1
        return self.ctx.powmod(self._as_cipher(base), _native.ints_to_limbs(exps, width))

    def invert_n2(self, a):
        return self.ctx.invert(self._as_cipher(a))

    # ---- the same five functions on device-resident arrays (DeviceArray in, DeviceArray out) --------------
    def upload_plain(self, ints_or_limbs):
        return DeviceArray.from_host(self.ctx, self._as_plain(ints_or_limbs))

    def upload_cipher(self, ints_or_limbs):
        return DeviceArray.from_host(self.ctx, self._as_cipher(ints_or_limbs))

    def raw_encrypt_dev(self, m, r):
        m = self.upload_plain([v % self.n for v in m] if not isinstance(m, np.ndarray) else m)
        r = self.upload_plain(r)
        out = DeviceArray(self.ctx, m.rows, self.ct_limbs)
        self.ctx.encrypt_dev(m.ptr, r.ptr, out.ptr, m.rows)
        self.ctx.sync()
        return out

    def obfuscate_dev(self, c, r):
        r = self.upload_plain(r)
        out = DeviceArray(self.ctx, c.rows, self.ct_limbs)
        self.ctx.obfuscate_dev(c.ptr, r.ptr, out.ptr, c.rows)
        self.ctx.sync()
        return out

    def raw_decrypt_dev(self, c):
        out = DeviceArray(self.ctx, c.rows, self.n_limbs)
        self.ctx.decrypt_dev(c.ptr, out.ptr, c.rows)
        self.ctx.sync()
        return out.to_host()

    def raw_add_dev(self, a, b):
        out = DeviceArray(self.ctx, a.rows, self.ct_limbs)
        self.ctx.mulmod_dev(a.ptr, b.ptr, out.ptr, a.rows)
        self.ctx.sync()
        return out

    def add_plain_dev(self, c, plaintexts):
        m = self.upload_plain([v % self.n for v in plaintexts] if not isinstance(plaintexts, np.ndarray) else plaintexts)
        out = DeviceArray(self.ctx, c.rows, self.ct_limbs)
        self.ctx.add_plain_dev(c.ptr, m.ptr, out.ptr, c.rows)
        self.ctx.sync()
        return out

    def powmod_dev(self, base, exps):
        exps = list(exps)
        bits = max(1, max(e.bit_length() for e in exps))
        width = (bits + 31) // 32
        e = DeviceArray.from_host(self.ctx, _native.ints_to_limbs(exps, width))
        out = DeviceArray(self.ctx, base.rows, self.ct_limbs)
        self.ctx.powmod_dev(base.ptr, e.ptr, width, bits, out.ptr, base.rows)
        self.ctx.sync()
        return out

    def raw_mul_dev(self, c, scalars):
        """Engine.raw_mul with the ciphertexts staying in HBM: the inverse branch (phe/paillier.py:745-749) is taken
        by inverting the whole vector on the device (3 modmuls per row) and selecting per row."""
        n, threshold = self.n, self.n - self.max_int
        neg = np.fromiter((s >= threshold for s in scalars), dtype=np.uint8, count=len(scalars))
        exps = [n - s if s >= threshold else s for s in scalars]
        base = c
        if neg.any():
            inv = DeviceArray(self.ctx, c.rows, self.ct_limbs)
            self.ctx.invert_dev(c.ptr, inv.ptr, c.rows)
            mask = DeviceArray.from_host(self.ctx, neg, dtype=np.uint8)
            base = DeviceArray(self.ctx, c.rows, self.ct_limbs)
            self.ctx.select_rows_dev(c.ptr, inv.ptr, mask.ptr, base.ptr, self.ct_limbs, c.rows)
            self.ctx.sync()
        return self.powmod_dev(base, exps)
