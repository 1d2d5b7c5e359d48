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
import threading

import numpy as np

from . import _native
from ._device import DeviceArray


def default_device():
    """One process per GPU: LOCAL_RANK picks the device unless PHE_HIP_DEVICE overrides it.  A single process that fans out
    over several devices (PHE_HIP_DEVICES, phe/fleet.py) keeps its ordinary engine on the first of them."""
    v = os.environ.get("PHE_HIP_DEVICE")
    if v is not None and v.strip() != "":
        return int(v)
    from . import fleet
    devices = fleet.configured_devices()
    if devices:
        return devices[0]
    v = os.environ.get("LOCAL_RANK")
    if v is not None and v.strip() != "":
        return int(v)
    return 0


def random_lt_n(n, count):
    """`count` cryptographically random integers in [1, n) as Python ints (see random_lt_n_limbs)."""
    limbs = (n.bit_length() + 31) // 32
    return _native.limbs_to_ints(random_lt_n_limbs(n, count, limbs)) if count else []


_urandom = None


def _urandom_into(view):
    """Fill a writable buffer from the kernel CSPRNG (/dev/urandom, what os.urandom and random.SystemRandom read)
    without an intermediate bytes object."""
    global _urandom
    if _urandom is None:
        _urandom = open("/dev/urandom", "rb", buffering=0)
    mv = memoryview(view).cast("B")
    got = 0
    while got < len(mv):
        k = _urandom.readinto(mv[got:])
        if not k:
            raise OSError("/dev/urandom returned no data")
        got += k


def random_lt_n_limbs(n, count, limbs, out=None):
    """`count` cryptographically random integers in [1, n) as a (count, limbs) uint32 array — the bulk form of
    PaillierPublicKey.get_random_lt_n (phe/paillier.py:141-143, random.SystemRandom().randrange(1, n)): same
    source (the kernel CSPRNG), same distribution (rejection sampling on bit_length(n-1) bits), vectorised with
    numpy.  `out`: optional (count, limbs) uint32 array to fill (saves the first-touch cost of a fresh buffer)."""
    k = (n - 1).bit_length()
    top_limb, top_bits = (k - 1) // 32, k - 32 * ((k - 1) // 32)
    top_mask = np.uint32((1 << top_bits) - 1) if top_bits < 32 else np.uint32(0xffffffff)
    n_arr = _native.int_to_limbs(n, limbs)
    if out is None:
        out = np.empty((count, limbs), dtype=np.uint32)
    if count == 0:
        return out
    _urandom_into(out)
    rows = None                                  # None = every row (column views, no gather)
    while True:
        sub = out if rows is None else out[rows]
        if top_bits < 32:
            sub[:, top_limb] &= top_mask
        if top_limb + 1 < limbs:
            sub[:, top_limb + 1:] = 0
        # rows that are >= n (lexicographic compare from the most significant limb down) or zero are redrawn
        lt = np.zeros(len(sub), dtype=bool)
        undecided = np.ones(len(sub), dtype=bool)
        for j in range(top_limb, -1, -1):
            col = sub[:, j]
            lt |= undecided & (col < n_arr[j])
            undecided &= (col == n_arr[j])
            if not undecided.any():
                break
        maybe_zero = lt & (sub[:, top_limb] == 0)
        if maybe_zero.any():
            idx = np.nonzero(maybe_zero)[0]
            lt[idx] = sub[idx].any(axis=1)
        if rows is not None:
            out[rows] = sub
        bad = np.nonzero(~lt)[0]
        if len(bad) == 0:
            return out
        rows = bad if rows is None else rows[bad]
        fresh = np.empty((len(rows), limbs), dtype=np.uint32)
        _urandom_into(fresh)
        out[rows] = fresh


class ObfuscatorPool:
    """Obfuscators r^n mod n^2 made ahead of time, each handed out ONCE (two ciphertexts sharing one reveal m1 - m2).
    The pool is an object of its own with its own lock because it outlives the engine it was filled through: when a key
    pair's private engine takes over from the public one (keys.PaillierPrivateKey._get_engine) both engines hold the SAME
    pool, so a thread still inside the old engine and a thread inside the new one cannot be handed the same rows.
    blocks: [DeviceArray, rows already used]; form: "pair" (rows in the engine's pair form) or "u32"."""

    def __init__(self):
        self.lock = threading.RLock()
        self.blocks = []

    def available(self):
        with self.lock:
            return sum(b.rows - used for b, used in self.blocks)

    def add(self, block):
        with self.lock:
            self.blocks.append([block, 0])

    def clear(self):
        with self.lock:
            self.blocks = []

    def take(self, count):
        """views of `count` unused rows (a list of DeviceArray views), or None if the pool is short; consumed"""
        with self.lock:
            if count <= 0 or sum(b.rows - used for b, used in self.blocks) < count:
                return None
            parts, need = [], count
            while need:
                block, used = self.blocks[0]
                k = min(need, block.rows - used)
                parts.append(block.rows_view(used, used + k))
                self.blocks[0][1] = used + k
                need -= k
                if self.blocks[0][1] == block.rows:
                    self.blocks.pop(0)
            return parts


# Every public Engine method runs under the engine's re-entrant lock (see _native.serialised).  A key's engine owns
# shared state: the obfuscator pool, whose entries must never be handed out twice (two ciphertexts sharing r^n reveal
# m1 - m2), the host staging buffers, the native context's window tables and launch stream.
@_native.serialised
class Engine:
    def __init__(self, n, p=None, q=None, hp=None, hq=None, p_inverse=None, device=None):
        self._lock = threading.RLock()
        self.n = n
        self.nsquare = n * n
        self.max_int = n // 3 - 1
        self.device = default_device() if device is None else device
        self.ctx = _native.Context(n, p, q, hp, hq, p_inverse, device=self.device)
        self.n_limbs = self.ctx.n_limbs
        self.ct_limbs = self.ctx.ct_limbs
        self._obf = ObfuscatorPool()

    # ---- reusable host staging (a fresh 32 MB numpy buffer costs more in page faults than the kernel it feeds) ---
    def scratch(self, name, rows, cols):
        """A (rows, cols) uint32 view of a grow-only buffer owned by this engine; contents are transient: valid
        until the next scratch(name, ...) call."""
        if not hasattr(self, "_scratch"):
            self._scratch = {}
        need = rows * cols
        buf = self._scratch.get(name)
        if buf is None or buf.size < need:
            buf = np.empty(max(need, 1), dtype=np.uint32)
            self._scratch[name] = buf
        return buf[:need].reshape(rows, cols)

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
    def owner_encrypt(self):
        """True when this engine holds the private key and the library offers the CRT form of raw_encrypt for the key
        width (include/phe_hip.h phe_hip_encrypt_owner_dev): r^n mod n^2 from r^n mod p^2 and r^n mod q^2 — about half
        the multiply-adds, the same bits.  Every encryption path of the engine takes it then."""
        ok = self.__dict__.get("_owner_ok")
        if ok is None:
            probe = getattr(self.ctx, "owner_encrypt_offered", None)
            ok = self._owner_ok = bool(probe()) if probe and os.environ.get("PHE_HIP_OWNER_ENCRYPT", "1") != "0" else False
        return ok

    def _encrypt_dev(self, m_ptr, r_ptr, c_ptr, rows, stream=0):
        fn = self.ctx.encrypt_owner_dev if self.owner_encrypt() else self.ctx.encrypt_dev
        fn(m_ptr, r_ptr, c_ptr, rows, stream)

    def raw_encrypt(self, m, r):
        """(1 + n*m) * r^n mod n^2 per row.  m is reduced mod n on the way in (value-identical to the
        reference's `% nsquare`, which lets m = n, n+1 wrap to 0, 1: phe/tests/paillier_test.py:114-126)."""
        if not isinstance(m, np.ndarray):
            m = self.plain_limbs([v % self.n for v in m])
        if self.owner_encrypt():
            return self.ctx.encrypt_owner(m, self._as_plain(r))
        return self.ctx.encrypt(m, self._as_plain(r))

    def obfuscate(self, c, r):
        c, r = self._as_cipher(c), self._as_plain(r)
        if self.owner_encrypt():
            # r^n through the key owner's CRT halves (raw_encrypt of the plaintext 0), then one product: same bits
            rn = self.ctx.encrypt_owner(np.zeros((r.shape[0], self.n_limbs), dtype=np.uint32), r)
            return self.ctx.mulmod(c, rn)
        return self.ctx.obfuscate(c, r)

    def _obfuscate_dev(self, c_ptr, r_ptr, out_ptr, rows, stream=0):
        """c * r^n on resident rows: the fused / composed public kernels, or — for a key owner — r^n from the CRT halves
        into `out`, then the product with c in place"""
        if not self.owner_encrypt():
            self.ctx.obfuscate_dev(c_ptr, r_ptr, out_ptr, rows, stream)
            return
        zeros = self.__dict__.get("_zero_plain")
        if zeros is None or zeros.rows < rows:
            zeros = self._zero_plain = DeviceArray.from_host(self.ctx, np.zeros((max(rows, 1 << 16), self.n_limbs), dtype=np.uint32))
        self.ctx.encrypt_owner_dev(zeros.ptr, r_ptr, out_ptr, rows, stream)
        self.ctx.mulmod_dev(c_ptr, out_ptr, out_ptr, rows, stream)

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

    @staticmethod
    def _mag_limbs(mag):
        mag = np.ascontiguousarray(mag, dtype=np.uint64)
        bits = int(mag.max()).bit_length() if len(mag) else 1
        return mag.view(np.uint32).reshape(len(mag), 2)[:, :1 if bits <= 32 else 2].copy(), max(bits, 1)

    def raw_mul_signed(self, c, mag, neg):
        """raw_mul for scalars given as sign + 64-bit magnitude (the array form of the codec): a negative scalar -v
        is the residue n - v >= n - max_int, so it takes the inverse branch with exponent v (phe/paillier.py:745-749)."""
        c = self._as_cipher(c)
        exps, _ = self._mag_limbs(mag)
        base = c
        if neg.any():
            idx = np.nonzero(neg)[0]
            base = c.copy()
            base[idx] = self.ctx.invert(np.ascontiguousarray(c[idx]))
        return self.ctx.powmod(base, exps)

    def _inverted_where(self, c, neg):
        """a vector equal to c with the rows where `neg` holds replaced by invert(c_i, n^2): the base of the negative-scalar branch
        of _raw_mul (phe/paillier.py:745-749).  Few negative rows (under a quarter): only those are inverted — gathered, inverted
        as a batch of their own (the simultaneous-inversion tree costs 3 products per row it is given), scattered into a copy;
        otherwise the whole vector is inverted and selected per row.  ZeroDivisionError carries the row's index in c."""
        neg = np.asarray(neg, dtype=bool)
        count = int(neg.sum())
        if count * 4 < c.rows and hasattr(self.ctx, "gather_rows_dev"):
            idx = np.nonzero(neg)[0].astype(np.uint32)
            idx_d = DeviceArray.from_host(self.ctx, idx)
            sub = DeviceArray(self.ctx, count, self.ct_limbs)
            self.ctx.gather_rows_dev(c.ptr, c.rows, idx_d.ptr, sub.ptr, self.ct_limbs, count)
            inv = DeviceArray(self.ctx, count, self.ct_limbs)
            try:
                self.ctx.invert_dev(sub.ptr, inv.ptr, count)
            except ZeroDivisionError as e:
                if getattr(e, "bad_index", None) is not None and e.bad_index < count:
                    e.bad_index = int(idx[e.bad_index])
                raise
            base = c.copy()
            self.ctx.scatter_rows_dev(inv.ptr, idx_d.ptr, base.ptr, base.rows, self.ct_limbs, count)
            self.ctx.sync()
            return base
        inv = DeviceArray(self.ctx, c.rows, self.ct_limbs)
        self.ctx.invert_dev(c.ptr, inv.ptr, c.rows)
        mask = DeviceArray.from_host(self.ctx, neg.astype(np.uint8), dtype=np.uint8)
        base = DeviceArray(self.ctx, c.rows, self.ct_limbs)
        self.ctx.select_rows_dev(c.ptr, inv.ptr, mask.ptr, base.ptr, self.ct_limbs, c.rows)
        self.ctx.sync()
        return base

    def raw_mul_signed_dev(self, c, mag, neg):
        exps, bits = self._mag_limbs(mag)
        base = c
        if neg.any():
            base = self._inverted_where(c, neg)
        e = DeviceArray.from_host(self.ctx, exps)
        out = DeviceArray(self.ctx, base.rows, self.ct_limbs)
        self.ctx.powmod_dev(base.ptr, e.ptr, exps.shape[1], bits, out.ptr, base.rows)
        self.ctx.sync()
        return out

    def __del__(self):
        try:
            st = self.__dict__.get("_stream")
            if st:
                self.ctx.stream_destroy(st)
        except Exception:
            pass

    @staticmethod
    def shifted_limbs(mag, shift_bits):
        """(mag[i] << shift_bits[i]) as little-endian uint32 limb rows, without a Python integer per element: the
        exponents k_i * BASE^delta_i of an aligned dot product (mag: uint64 magnitudes, shift_bits: int64 >= 0).
        Returns (limbs (batch, width), largest bit length)."""
        mag = np.ascontiguousarray(mag, dtype=np.uint64).reshape(-1)
        sh = np.ascontiguousarray(shift_bits, dtype=np.int64).reshape(-1)
        count = len(mag)
        top = int(mag.max()) if count else 0
        if top == 0:
            return np.zeros((count, 1), dtype=np.uint32), 1
        nz = mag != 0
        sh = np.where(nz, sh, 0)              # a zero stays zero whatever its shift: it must not size or index the rows
        # exact bit length of every magnitude: float64 log2 can be off by one near powers of two, so fix it up
        bl = np.zeros(count, dtype=np.int64)
        est = np.floor(np.log2(mag[nz].astype(np.float64))).astype(np.int64) + 1
        est = np.minimum(est, 64)
        m_nz = mag[nz]
        too_big = (est > 1) & ((m_nz >> (est - 1).astype(np.uint64)) == 0)
        est[too_big] -= 1
        too_small = (est < 64) & ((m_nz >> np.minimum(est, 63).astype(np.uint64)) != 0)
        est[too_small] += 1
        bl[nz] = est
        bits = int((bl + sh).max())
        width = max(1, (bits + 31) // 32)
        out = np.zeros((count, width + 3), dtype=np.uint32)       # 3 spill words, cut off below
        word = sh >> 5
        bo = (sh & 31).astype(np.uint64)
        lo = mag << bo                                            # low 64 bits of the shifted value
        hi = np.where(bo == 0, np.uint64(0), mag >> ((np.uint64(64) - bo) & np.uint64(63)))   # the bits shifted out
        rows = np.arange(count)
        out[rows, word] = (lo & np.uint64(0xffffffff)).astype(np.uint32)
        out[rows, word + 1] = (lo >> np.uint64(32)).astype(np.uint32)
        out[rows, word + 2] = hi.astype(np.uint32)
        return np.ascontiguousarray(out[:, :width]), max(bits, 1)

    @staticmethod
    def _exp_limbs(exps):
        """exponents (uint64 array or Python ints) -> (batch, width) uint32 limbs and the largest bit length"""
        if isinstance(exps, np.ndarray):
            return Engine._mag_limbs(exps)
        exps = list(exps)
        bits = max(1, max((e.bit_length() for e in exps), default=1))
        return _native.ints_to_limbs(exps, (bits + 31) // 32), bits

    def raw_dot(self, c, exps, neg):
        """prod_i b_i^e_i mod n^2 with b_i = c_i, or invert(c_i, n^2) where neg[i] — the ciphertext of sum_i k_i * x_i
        exactly as a chain of _raw_mul (phe/paillier.py:745-751: scalars at or above n - max_int take the inverted
        base with exponent n - k) and _raw_add (:705-719) leaves it: the product of canonical residues does not
        depend on the order of the factors.  One k_multiexp_split launch plus the k_mulmod tree over its chunk
        products (include/phe_hip.h phe_hip_multiexp).  c: host limb array or DeviceArray.  Returns a Python int."""
        limbs, bits = exps if isinstance(exps, tuple) else self._exp_limbs(exps)
        neg = np.asarray(neg, dtype=bool)
        if isinstance(c, DeviceArray) and self.pair_form() and c.cols == self.pair_form() != self.ct_limbs:
            # rows resident in the pair form, no negative scalar (the caller checked): the multi-exponentiation reads them as they
            # are — no conversion of every ciphertext into the form (include/phe_hip.h phe_hip_pair_multiexp_rows_dev)
            if neg.any():
                raise ValueError("pair-form rows take non-negative scalars only")
            e = DeviceArray.from_host(self.ctx, limbs)
            out = DeviceArray(self.ctx, 1, self.ct_limbs)
            self.ctx.pair_multiexp_rows_dev(c.ptr, e.ptr, limbs.shape[1], bits, out.ptr, c.rows, 1)
            self.ctx.sync()
            return self.to_ints(out.to_host())[0]
        if isinstance(c, DeviceArray):
            base = c
            if neg.any():
                inv = DeviceArray(self.ctx, c.rows, self.ct_limbs)
                self.ctx.invert_dev(c.ptr, inv.ptr, c.rows)
                mask = DeviceArray.from_host(self.ctx, neg.astype(np.uint8), dtype=np.uint8)
                base = DeviceArray(self.ctx, c.rows, self.ct_limbs)
                self.ctx.select_rows_dev(c.ptr, inv.ptr, mask.ptr, base.ptr, self.ct_limbs, c.rows)
            e = DeviceArray.from_host(self.ctx, limbs)
            out = DeviceArray(self.ctx, 1, self.ct_limbs)
            self.ctx.multiexp_dev(base.ptr, e.ptr, limbs.shape[1], bits, out.ptr, c.rows)
            self.ctx.sync()
            return self.to_ints(out.to_host())[0]
        base = self._as_cipher(c)
        if neg.any():
            idx = np.nonzero(neg)[0]
            base = base.copy()
            base[idx] = self.ctx.invert(np.ascontiguousarray(base[idx]))
        return self.to_ints(self.ctx.multiexp(base, limbs))[0]

    def raw_matvec(self, c, exps, neg):
        """rows of raw_dot over the same ciphertexts: out[r] = prod_i b_i^exps[r][i] mod n^2, b_i = c_i or
        invert(c_i, n^2) where neg[r][i] (include/phe_hip.h phe_hip_multiexp_rows_dev: the tables of a chunk of
        ciphertexts serve a block of rows; the vector is inverted once with the simultaneous-inversion tree when any
        entry is negative).  exps: (rows, batch) uint64 array or list of rows of Python ints; neg: (rows, batch) bool.
        c: host limb array or DeviceArray; returns the same kind, (rows, ct_limbs)."""
        neg = np.asarray(neg, dtype=bool)
        rows, batch = neg.shape
        if isinstance(exps, tuple):                              # (limb rows (rows*batch, width), bits): shifted_limbs
            limbs, bits = exps[0].reshape(rows, batch, -1), exps[1]
        elif isinstance(exps, np.ndarray):
            e64 = np.ascontiguousarray(exps, dtype=np.uint64)
            bits = int(e64.max()).bit_length() if e64.size else 1
            limbs = e64.view(np.uint32).reshape(rows, batch, 2)[:, :, :1 if bits <= 32 else 2].copy()
        else:
            bits = max(1, max((v.bit_length() for row in exps for v in row), default=1))
            width = (bits + 31) // 32
            limbs = np.stack([_native.ints_to_limbs(list(row), width) for row in exps]) if rows else np.zeros((0, batch, width), np.uint32)
        bits = max(bits, 1)
        on_dev = isinstance(c, DeviceArray)
        any_neg = bool(neg.any())
        if on_dev and self.pair_form() and c.cols == self.pair_form() != self.ct_limbs:
            if any_neg:
                raise ValueError("pair-form rows take non-negative scalars only")
            e = DeviceArray.from_host(self.ctx, limbs.reshape(rows * batch, -1))
            out = DeviceArray(self.ctx, rows, self.ct_limbs)
            self.ctx.pair_multiexp_rows_dev(c.ptr, e.ptr, limbs.shape[2], bits, out.ptr, batch, rows)
            self.ctx.sync()
            return out
        info = self.ctx.info()
        if not (info.get("emulated") or info.get("engine_pub") == "split"):
            # keys without a split geometry: row by row on the single-row entry point
            out = [self.raw_dot(c, (np.ascontiguousarray(limbs[r]), bits), neg[r]) for r in range(rows)]
            host = self.cipher_limbs(out)
            return DeviceArray.from_host(self.ctx, host) if on_dev else host
        if on_dev:
            inv = None
            if any_neg:
                inv = DeviceArray(self.ctx, c.rows, self.ct_limbs)
                self.ctx.invert_dev(c.ptr, inv.ptr, c.rows)
            e = DeviceArray.from_host(self.ctx, limbs.reshape(rows * batch, -1))
            mask = DeviceArray.from_host(self.ctx, neg.astype(np.uint8).reshape(-1), dtype=np.uint8) if any_neg else None
            out = DeviceArray(self.ctx, rows, self.ct_limbs)
            self.ctx.multiexp_rows_dev(c.ptr, inv.ptr if inv else None, e.ptr, mask.ptr if mask else None, limbs.shape[2], bits,
                                       out.ptr, batch, rows)
            self.ctx.sync()
            return out
        base = self._as_cipher(c)
        inv = self.ctx.invert(base) if any_neg else None
        return self.ctx.multiexp_rows(base, inv, limbs, neg.astype(np.uint8) if any_neg else None)

    def has_split_engine(self):
        info = self.ctx.info()
        return bool(info.get("emulated") or info.get("engine_pub") == "split")

    def raw_matvec_csr(self, c, row_ptr, cols, exps, neg, rows):
        """out[r] = prod over the entries of row r of b[col]^exp (b = c[col], or its inverse where neg[entry]): the
        table-lookup form of raw_matvec (include/phe_hip.h phe_hip_multiexp_csr_dev) — tables of the whole vector
        once, then one ladder per row over its entries only.  row_ptr / cols: CSR (None, None = dense rows);
        exps: (limb rows (entries, width), bits) as shifted_limbs returns them; neg: (entries,) bool or None.
        Rows are visited longest first.  c: host limb array or DeviceArray; returns the same kind."""
        limbs, bits = exps
        any_neg = neg is not None and bool(np.asarray(neg).any())
        order = None
        if row_ptr is not None:
            row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64)
            counts = np.diff(row_ptr.astype(np.int64))
            order = np.argsort(-counts, kind="stable").astype(np.uint32)
        if isinstance(c, DeviceArray):
            inv = None
            if any_neg:
                inv = DeviceArray(self.ctx, c.rows, self.ct_limbs)
                self.ctx.invert_dev(c.ptr, inv.ptr, c.rows)
            dev = lambda a, dt: DeviceArray.from_host(self.ctx, np.ascontiguousarray(a, dtype=dt).reshape(-1), dtype=dt)
            e = DeviceArray.from_host(self.ctx, limbs)
            rp = DeviceArray.from_host(self.ctx, row_ptr.view(np.uint32).reshape(-1, 2)) if row_ptr is not None else None
            cl = DeviceArray.from_host(self.ctx, np.ascontiguousarray(cols, dtype=np.uint32)) if cols is not None else None
            ng = dev(np.asarray(neg).astype(np.uint8), np.uint8) if any_neg else None
            od = DeviceArray.from_host(self.ctx, order) if order is not None else None
            out = DeviceArray(self.ctx, rows, self.ct_limbs)
            ptr = lambda a: a.ptr if a is not None else None
            self.ctx.multiexp_csr_dev(c.ptr, ptr(inv), c.rows, ptr(rp), ptr(cl), e.ptr, ptr(ng), limbs.shape[1], bits, ptr(od),
                                      out.ptr, rows)
            self.ctx.sync()
            return out
        base = self._as_cipher(c)
        inv = self.ctx.invert(base) if any_neg else None
        return self.ctx.multiexp_csr(base, inv, row_ptr, cols, limbs, np.asarray(neg).astype(np.uint8) if any_neg else None,
                                     order, rows)

    # ---- the decimal wire format of ciphertext vectors (docs/serialisation.rst:24-43; str(int) / int(str) per
    #      element in the reference) ---------------------------------------------------------------------------
    def decimal_strings(self, c):
        """ciphertext rows (host limb array or DeviceArray) -> list of str, exactly [str(int) ...]: the base
        conversion runs on the device (csrc/radix_conv.h), the host only strips the '0' padding"""
        if isinstance(c, DeviceArray):
            digits = self.ctx.to_decimal_dev(c.ptr, c.cols, c.rows)
        else:
            digits = self.ctx.to_decimal(c)
        width = digits.shape[1]
        raw = digits.tobytes()
        return [raw[i:i + width].lstrip(b"0").decode("ascii") or "0" for i in range(0, len(raw), width)]

    def limbs_from_decimal(self, strings, words=None):
        """list of decimal strings -> (rows, words) limb array (int(str) per element in the reference, then the
        reduction the kernels expect is the caller's business).  ValueError for anything int() would reject here
        (signs, spaces and underscores are not part of the wire format) or a value beyond `words` words."""
        words = words or self.ct_limbs
        if not strings:
            return np.zeros((0, words), dtype=np.uint32)
        width = max(1, max(len(s) for s in strings))
        raw = b"".join(s.encode("ascii").rjust(width, b"0") for s in strings)
        digits = np.frombuffer(raw, dtype=np.uint8).reshape(len(strings), width)
        return self.ctx.from_decimal(digits, words)

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
        self._encrypt_dev(m.ptr, r.ptr, out.ptr, m.rows)
        self.ctx.sync()
        return out

    def _launch_stream(self):
        """a non-blocking stream for pipelined launches (created on first use; 0 = NULL stream if the backend has none)"""
        st = self.__dict__.get("_stream")
        if st is None:
            st = self.ctx.stream_create() if hasattr(self.ctx, "stream_create") else 0
            self._stream = st
        return st

    def raw_encrypt_fresh(self, m, device):
        """raw_encrypt with freshly drawn obfuscators, in chunks: the draw (the kernel CSPRNG, host side) and the upload
        of chunk k+1 overlap the kernel of chunk k — the kernels are queued on a non-blocking stream, so the blocking
        copies (NULL stream) do not wait for them; the first chunk is half size to shorten the exposed prologue.
        m: (count, n_limbs) plaintext limbs.  Returns a DeviceArray or a host array."""
        count = m.shape[0]
        st = self._launch_stream()
        out = DeviceArray(self.ctx, count, self.ct_limbs)
        host = None if device else np.empty((count, self.ct_limbs), dtype=np.uint32)
        keep = []                                         # operand buffers stay alive until the final sync
        lo, chunk, prev = 0, 1 << 15, None                # chunks are multiples of the groups in flight (32768 at 2048 bits)
        while lo < count:
            hi = min(count, lo + chunk)
            r = random_lt_n_limbs(self.n, hi - lo, self.n_limbs, out=self.scratch("r", hi - lo, self.n_limbs))
            m_d = DeviceArray.from_host(self.ctx, m[lo:hi])
            r_d = DeviceArray.from_host(self.ctx, r)      # synchronous copy: the scratch buffer is free again
            keep += [m_d, r_d]
            if host is not None and st:
                self.ctx.sync(st)                         # the previous chunk is complete (its successor starts right away)
            self._encrypt_dev(m_d.ptr, r_d.ptr, out.rows_view(lo, hi).ptr, hi - lo, st)
            if host is not None and st and prev is not None:
                self.ctx.d2h(host[prev[0]:prev[1]], out.rows_view(*prev).ptr)     # download under this chunk's kernel
            prev = (lo, hi)
            lo, chunk = hi, 1 << 16
        self.ctx.sync(st)
        if host is None:
            return out
        if st and prev is not None:
            self.ctx.d2h(host[prev[0]:prev[1]], out.rows_view(*prev).ptr)
            return host
        return out.to_host()

    # ---- offline / online split: obfuscators r^n mod n^2 made ahead of time -------------------------------------------
    # r^n is the expensive factor of an encryption and does not depend on the plaintext (phe/paillier.py:137 draws r
    # and exponentiates at encryption time).  A pool of them, each used ONCE, turns the online part of encrypt_batch /
    # obfuscate into one product per element: c = (1 + n m) * r^n — the same value raw_encrypt(m, r) returns for that r.
    def fill_obfuscator_pool(self, count):
        """draw `count` fresh r in [1, n) and keep r^n mod n^2 in HBM (raw_encrypt of the plaintext 0: 1 + n*0 = 1) — as
        rows of the engine's pair form where it has one: the online part of an encryption is then one exit from that form
        with the plaintext folded in (Engine.encrypt_from_obfuscators)"""
        if count <= 0:
            return self.obfuscators_available()
        zeros = np.zeros((count, self.n_limbs), dtype=np.uint32)
        block = self.raw_encrypt_fresh(zeros, device=True)
        if self.pair_form():
            block = self.to_pair_dev(block)
        self._obf.add(block)
        return self.obfuscators_available()

    def clear_obfuscator_pool(self):
        """drop the obfuscators made ahead of time (nothing was handed out twice; the unused ones are simply forgotten)"""
        self._obf.clear()

    def obfuscators_available(self):
        return self._obf.available()

    def take_pool_rows(self, count):
        """`count` unused pool rows as ONE DeviceArray (a view where possible), or None; consumed.  The rows come in the form
        the engine works in NOW (pair_form(): pair rows, else ciphertext words): a block filled through another engine of the
        key pair, or under another PHE_HIP_PAIR_FORM setting, may be in the other form — every part is brought to one form
        BEFORE the parts are joined (a copy at the wrong row stride would corrupt the obfuscators)."""
        parts = self._obf.take(count)
        if parts is None:
            return None
        want = self.pair_form() or self.ct_limbs
        if want not in {part.cols for part in parts}:     # (an emulated / full-width engine never sees pair blocks)
            want = parts[0].cols

        def normal(part):
            if part.cols == want:
                return part
            return self.from_pair_dev(part) if want == self.ct_limbs else self.to_pair_dev(part)
        parts = [normal(part) for part in parts]
        if len(parts) == 1:
            return parts[0]
        out = DeviceArray(self.ctx, count, want)
        lo = 0
        for part in parts:
            self.ctx.d2d(out.ptr + lo * want * 4, part.ptr, part.nbytes)
            lo += part.rows
        self.ctx.sync()
        return out

    _take_pool_rows = take_pool_rows

    def take_obfuscators(self, count):
        """`count` unused obfuscators r^n mod n^2 as plain ciphertext rows (DeviceArray of ct_limbs words), or None if
        the pool is short; they are consumed: nothing is handed out twice"""
        rows = self.take_pool_rows(count)
        if rows is None or rows.cols == self.ct_limbs:
            return rows
        return self.from_pair_dev(rows)

    def encrypt_from_obfuscators(self, plaintexts):
        """raw_encrypt(m_i, r_i) for pooled r_i (each used once): (1 + n*m_i) * r_i^n mod n^2 as a DeviceArray, or None
        if the pool is short.  plaintexts: (count, n_limbs) limb rows or Python ints."""
        if not isinstance(plaintexts, np.ndarray):
            plaintexts = self.plain_limbs([v % self.n for v in plaintexts])
        rows = self.take_pool_rows(plaintexts.shape[0])
        if rows is None:
            return None
        if rows.cols == self.ct_limbs:
            return self.add_plain_dev(rows, plaintexts)
        return self.from_pair_dev(rows, plaintexts)

    def peek_obfuscators(self, count):
        """the next `count` unused obfuscators r^n mod n^2 as Python ints, WITHOUT consuming them (tests, diagnostics)"""
        out = []
        with self._obf.lock:
            for block, used in self._obf.blocks:
                k = min(count - len(out), block.rows - used)
                if k <= 0:
                    break
                view = block.rows_view(used, used + k)
                if view.cols != self.ct_limbs:
                    view = self.from_pair_dev(view)
                out += self.to_ints(view.to_host())
        return out

    # ---- resident rows in the pair form (include/phe_hip.h "pair form") ---------------------------------------------------
    def pair_form(self):
        """words of a pair-form row (0: not offered — no split-modulus engine, an emulated backend, PHE_HIP_PAIR_FORM=0)"""
        w = self.__dict__.get("_pair_words")
        if w is None:
            probe = getattr(self.ctx, "pair_words", None)
            w = self._pair_words = int(probe()) if probe and os.environ.get("PHE_HIP_PAIR_FORM", "1") != "0" else 0
        return w

    def to_pair_dev(self, c):
        out = DeviceArray(self.ctx, c.rows, self.pair_form())
        self.ctx.to_pair_dev(c.ptr, out.ptr, c.rows)
        self.ctx.sync()
        return out

    def from_pair_dev(self, pair, plaintexts=None):
        """pair rows -> canonical ciphertext rows; with plaintexts (limb rows / DeviceArray) the residue of x * (1 + n*m)"""
        m = None
        if plaintexts is not None:
            m = plaintexts if isinstance(plaintexts, DeviceArray) else self.upload_plain(plaintexts)
        out = DeviceArray(self.ctx, pair.rows, self.ct_limbs)
        self.ctx.from_pair_dev(pair.ptr, m.ptr if m is not None else None, out.ptr, pair.rows)
        self.ctx.sync()
        return out

    def pair_mul_dev(self, a, b):
        """row-wise product of two pair-form vectors (b with one row: that row for every a): one homomorphic addition"""
        out = DeviceArray(self.ctx, a.rows, a.cols)
        self.ctx.pair_mul_dev(a.ptr, b.ptr, b.rows == 1 and a.rows != 1, out.ptr, a.rows)
        self.ctx.sync()
        return out

    def pair_mul_scalars_dev(self, pair, mag):
        """row-wise power a[i]^mag[i] of a pair-form vector, the result in pair form: _raw_mul by non-negative scalars on resident
        rows, without the conversion in and the exit of powmod_dev"""
        exps, bits = self._mag_limbs(mag)
        e = DeviceArray.from_host(self.ctx, exps)
        out = DeviceArray(self.ctx, pair.rows, pair.cols)
        self.ctx.pair_powmod_dev(pair.ptr, e.ptr, exps.shape[1], bits, out.ptr, pair.rows)
        self.ctx.sync()
        return out

    def pair_reduce_dev(self, pair):
        """the product of all rows of a pair-form vector as one pair row (the tree of EncryptedVector.sum, one call)"""
        out = DeviceArray(self.ctx, 1, pair.cols)
        self.ctx.pair_reduce_dev(pair.ptr, pair.rows, out.ptr)
        self.ctx.sync()
        return out

    def obfuscate_fresh_dev(self, c, rows=None):
        """obfuscate_dev with freshly drawn obfuscators, chunked like raw_encrypt_fresh (draw and upload of the next
        chunk under the kernels of the current one).  rows: indices that get a fresh r (None = all); the others get
        r = 1 (c * 1^n = c, still canonical)."""
        count = c.rows
        st = self._launch_stream()
        need = None
        if rows is not None:
            need = np.zeros(count, dtype=bool)
            need[rows] = True
        out = DeviceArray(self.ctx, count, self.ct_limbs)
        keep = []
        lo, chunk = 0, 1 << 15
        while lo < count:
            hi = min(count, lo + chunk)
            if need is None:
                r = random_lt_n_limbs(self.n, hi - lo, self.n_limbs, out=self.scratch("r", hi - lo, self.n_limbs))
            else:
                r = self.scratch("r", hi - lo, self.n_limbs)
                r[:] = 0
                r[:, 0] = 1
                sel = np.nonzero(need[lo:hi])[0]
                if len(sel):
                    r[sel] = random_lt_n_limbs(self.n, len(sel), self.n_limbs)
            r_d = DeviceArray.from_host(self.ctx, r)
            keep.append(r_d)
            self._obfuscate_dev(c.rows_view(lo, hi).ptr, r_d.ptr, out.rows_view(lo, hi).ptr, hi - lo, st)
            lo, chunk = hi, 1 << 16
        self.ctx.sync(st)
        return out

    def obfuscate_dev(self, c, r):
        r = self.upload_plain(r)
        out = DeviceArray(self.ctx, c.rows, self.ct_limbs)
        self._obfuscate_dev(c.ptr, r.ptr, out.ptr, c.rows)
        self.ctx.sync()
        return out

    def raw_decrypt_dev(self, c):
        out = DeviceArray(self.ctx, c.rows, self.n_limbs)
        self.ctx.decrypt_dev(c.ptr, out.ptr, c.rows)
        self.ctx.sync()
        return out.to_host()

    def raw_decrypt_dev_chunks(self, c, chunk=1 << 16):
        """raw_decrypt of a resident vector as a generator of (lo, hi, plaintext limbs on the host): while the caller
        works on one chunk (download + decoding), the kernels of the next one run — they are queued on the engine's
        non-blocking stream right after the finished chunk is awaited, and the blocking download does not wait for them.
        (A generator: the engine's lock is taken around each step, never across a yield — see _native.serialised.)"""
        with self._lock:
            st = self._launch_stream()
        rows = c.rows
        if rows <= chunk or not st:
            if rows:
                yield 0, rows, self.raw_decrypt_dev(c)
            return
        with self._lock:
            out = DeviceArray(self.ctx, rows, self.n_limbs)

        def launch(lo):
            hi = min(rows, lo + chunk)
            self.ctx.decrypt_dev(c.rows_view(lo, hi).ptr, out.rows_view(lo, hi).ptr, hi - lo, st)
            return hi
        try:
            with self._lock:
                lo, hi = 0, launch(0)
            while lo < rows:
                with self._lock:
                    self.ctx.sync(st)                           # chunk [lo, hi) is complete
                    nxt = launch(hi) if hi < rows else hi
                    host = out.rows_view(lo, hi).to_host()      # blocking copy on the NULL stream, under the next kernels
                yield lo, hi, host
                lo, hi = hi, nxt
        finally:
            # a consumer that stops early (e.g. OverflowError while decoding a row) drops the generator with the next
            # chunk's kernels still writing into `out`: wait for them before the block returns to the size-keyed pool
            self.ctx.sync(st)

    def raw_decrypt_host_chunks(self, c, chunk=1 << 16):
        """raw_decrypt_dev_chunks for a HOST limb array: the upload of chunk k+1 and the download + decoding of chunk k
        both happen while kernels run (uploads and downloads are blocking copies on the NULL stream, the kernels are
        queued on the engine's non-blocking stream)."""
        with self._lock:
            st = self._launch_stream()
        rows = c.shape[0]
        if rows <= chunk or not st:
            if rows:
                yield 0, rows, self.raw_decrypt(c)
            return
        bounds = [(lo, min(rows, lo + chunk)) for lo in range(0, rows, chunk)]
        inputs, outputs = {}, {}

        def stage(k):
            lo, hi = bounds[k]
            inputs[k] = DeviceArray.from_host(self.ctx, c[lo:hi])
            outputs[k] = DeviceArray(self.ctx, hi - lo, self.n_limbs)

        def launch(k):
            lo, hi = bounds[k]
            self.ctx.decrypt_dev(inputs[k].ptr, outputs[k].ptr, hi - lo, st)
        try:
            with self._lock:
                stage(0)
                launch(0)
            for k, (lo, hi) in enumerate(bounds):
                with self._lock:
                    if k + 1 < len(bounds):
                        stage(k + 1)                             # upload under the kernels of chunk k
                    self.ctx.sync(st)                            # chunk k is complete
                    if k + 1 < len(bounds):
                        launch(k + 1)
                    host = outputs.pop(k).to_host()              # download under chunk k+1
                    inputs.pop(k)
                yield lo, hi, host                               # the caller decodes under chunk k+1
        finally:
            self.ctx.sync(st)                                    # see raw_decrypt_dev_chunks: nothing in flight on release

    def raw_add_dev(self, a, b):
        out = DeviceArray(self.ctx, a.rows, self.ct_limbs)
        self.ctx.mulmod_dev(a.ptr, b.ptr, out.ptr, a.rows)
        self.ctx.sync()
        return out

    # ---- chains of additions on resident vectors: one Montgomery product per addition ("Montgomery debt") -------------
    # _raw_add is mulmod(a, b, n^2) (phe/paillier.py:705-719): on the device two Montgomery products, a*b/R and then *R^2/R.
    # A resident vector may instead hold x * R^-d mod n^2 for a vector-wide integer d (its debt): the product of two such
    # rows by ONE Montgomery product is (x_a x_b) * R^-(d_a + d_b + 1), and one product with the constant R^(d+1) mod n^2
    # gives the plain residues back when they are needed (download, decrypt, scalar multiplication, ...).  A chain of k
    # additions costs k + 1 products instead of 2k; what leaves the vector is the same canonical residue as before.
    def lazy_products(self):
        """True when the backend offers the one-product entry (include/phe_hip.h phe_hip_montmul_dev)"""
        ok = self.__dict__.get("_lazy_ok")
        if ok is None:
            ok = hasattr(self.ctx, "montmul_dev") and hasattr(self.ctx, "malloc")
            if ok:
                try:
                    self._mont_radix()
                except ValueError:                    # no full-width geometry for n^2 (keys above ~4170 bits)
                    ok = False
            self._lazy_ok = ok
        return ok

    def _mont_radix(self):
        R = self.__dict__.get("_mont_R")
        if R is None:
            R = self._mont_R = 1 << self.ctx.mont_radix_bits()
        return R

    def _mont_const_row(self, power):
        """the row R^power mod n^2 on the device (cached per power)"""
        rows = self.__dict__.setdefault("_mont_rows", {})
        row = rows.get(power)
        if row is None:
            if len(rows) > 256:
                rows.clear()
            row = rows[power] = DeviceArray.from_host(self.ctx, self.cipher_limbs([pow(self._mont_radix(), power, self.nsquare)]))
        return row

    def montmul_dev(self, a, b, stream=None):
        """row-wise a * b / R mod n^2 (canonical): the debts of the operands add up, plus one.  stream: queue the launch
        there and return without waiting (the caller synchronises that stream once, at the end of its chain; blocks freed in
        between are only reused by later launches of the SAME stream, i.e. in order); None: complete on return"""
        out = DeviceArray(self.ctx, a.rows, self.ct_limbs)
        self.ctx.montmul_dev(a.ptr, b.ptr, False, out.ptr, a.rows, stream or 0)
        if stream is None:
            self.ctx.sync()
        return out

    def scale_dev(self, a, power, stream=None):
        """row-wise a * R^power mod n^2 by one product with the constant R^(power + 1): power = d settles a debt of d,
        a negative power takes a row further into debt (to meet a partner's)"""
        const = self._mont_const_row(power + 1)                 # (uploaded with a blocking copy: complete before the launch)
        out = DeviceArray(self.ctx, a.rows, self.ct_limbs)
        self.ctx.montmul_dev(a.ptr, const.ptr, True, out.ptr, a.rows, stream or 0)
        if stream is None:
            self.ctx.sync()
        else:
            out._keep_const = const     # queued, not run yet: the constant lives as long as the result (the cache may evict it)
        return out

    def montmul_tree_dev(self, store, debt):
        """The pairwise product tree of EncryptedVector.sum over resident rows that hold x * R^-debt: ONE Montgomery product
        per node (a level turns debt d into 2d + 1; an unpaired last row is taken to the new debt by a product with a constant,
        written into the level's spare row), every level queued on the engine's launch stream, settled once at the root.
        Returns the root as a one-row DeviceArray of plain residues.  A public method: the whole tree runs under the engine's
        lock (stream creation and the constant-row cache are check-then-set)."""
        st = self._launch_stream() or None
        keep = [store]                                       # operands stay alive until the final synchronisation
        while store.rows > 1:
            half, odd = store.rows // 2, store.rows % 2
            merged = DeviceArray(self.ctx, half + odd, store.cols)
            self.ctx.montmul_dev(store.rows_view(0, half).ptr, store.rows_view(half, 2 * half).ptr, False, merged.ptr, half, st or 0)
            if odd:                                           # the row left over is taken to the new debt: * R^-(debt+1)
                const = self._mont_const_row(-(debt + 1) + 1)
                keep.append(const)                            # (the cache may evict the row while the queued kernel still reads it)
                self.ctx.montmul_dev(store.rows_view(2 * half, 2 * half + 1).ptr, const.ptr, True,
                                     merged.rows_view(half, half + 1).ptr, 1, st or 0)
            keep.append(merged)
            store, debt = merged, 2 * debt + 1
        root = self.scale_dev(store, debt, stream=st) if debt else store
        self.ctx.sync(st or 0)
        return root

    def add_plain_dev(self, c, plaintexts):
        m = self.upload_plain([v % self.n for v in plaintexts] if not isinstance(plaintexts, np.ndarray) else plaintexts)
        out = DeviceArray(self.ctx, c.rows, self.ct_limbs)
        self.ctx.add_plain_dev(c.ptr, m.ptr, out.ptr, c.rows)
        self.ctx.sync()
        return out

    def powmod_dev(self, base, exps):
        """exps: Python ints, or (limb rows (batch, width), largest bit length) as shifted_limbs returns them"""
        if isinstance(exps, tuple):
            limbs, bits = exps
        else:
            exps = list(exps)
            bits = max(1, max(e.bit_length() for e in exps))
            limbs = _native.ints_to_limbs(exps, (bits + 31) // 32)
        width = limbs.shape[1]
        e = DeviceArray.from_host(self.ctx, limbs)
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
        base = self._inverted_where(c, neg) if neg.any() else c
        return self.powmod_dev(base, exps)
