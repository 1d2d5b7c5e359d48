"""TEST INFRASTRUCTURE: an emulator-backed stand-in for phe._native.Context.

Lets the host-side Python package (python-paillier_amd/phe) be exercised on a CPU-only box: the calls
that would go through the C-ABI to the GPU are served by tests/emu/libphe_emu.so, i.e. by the same
device algorithm headers compiled for the host on the fiber wave emulator.  Installed only by tests
(monkeypatching phe._native.Context); the product has no such path.
"""
import numpy as np

from emu_lib import Emu

_emu = None


def emu():
    global _emu
    if _emu is None:
        _emu = Emu()
    return _emu


class EmuContext:
    def __init__(self, n, p=None, q=None, hp=None, hq=None, p_inverse=None, device=0, n_limbs=None):
        from phe import _native
        self._native = _native
        self.n = int(n)
        self.n_limbs = n_limbs or _native.limbs_for_bits(self.n.bit_length())
        self.ct_limbs = 2 * self.n_limbs
        self.device = device
        self.has_private = p is not None
        self._n_arr = _native.int_to_limbs(self.n, self.n_limbs)
        self._nsq_arr = _native.int_to_limbs(self.n * self.n, self.ct_limbs)
        if self.has_private:
            if p * q != n:
                raise ValueError("given public key does not match the given p and q.")
            pq = _native.limbs_for_bits(max(int(p).bit_length(), int(q).bit_length()))
            self._key = [_native.int_to_limbs(v, pq) for v in (p, q, hp, hq, p_inverse)]

    def close(self):
        pass

    def info(self):
        return {"n_limbs": self.n_limbs, "ct_limbs": self.ct_limbs, "emulated": True}

    def encrypt(self, m, r):
        if m.shape[0] == 0:
            return np.zeros((0, self.ct_limbs), np.uint32)
        return emu().encrypt(self._n_arr, np.ascontiguousarray(m), np.ascontiguousarray(r))

    def owner_encrypt_offered(self):
        return self.has_private and not self._wide() and self.n_limbs >= 2

    def encrypt_owner(self, m, r):
        if m.shape[0] == 0:
            return np.zeros((0, self.ct_limbs), np.uint32)
        out = emu().encrypt_owner(self._n_arr, *self._key, np.ascontiguousarray(m), np.ascontiguousarray(r))
        if out is None:
            raise ValueError("owner encryption is not offered for this key width")
        return out

    def obfuscate(self, c_in, r):
        if c_in.shape[0] == 0:
            return c_in.copy()
        return emu().obfuscate(self._n_arr, np.ascontiguousarray(c_in), np.ascontiguousarray(r))

    def decrypt(self, c):
        if not self.has_private:
            raise ValueError("decrypt needs a private-key context")
        if c.shape[0] == 0:
            return np.zeros((0, self.n_limbs), np.uint32)
        return emu().decrypt(*self._key, self.n_limbs, np.ascontiguousarray(c))

    def _wide(self):
        """n^2 beyond the widest full-width geometry (keys above ~4170 bits): products run on the pair form, as in the library"""
        return 2 * self.n.bit_length() + 4 > 16 * 18 * 29

    def mulmod(self, a, b):
        if a.shape[0] == 0:
            return a.copy()
        if self._wide():
            return emu().mulmod_n2_split(self._n_arr, np.ascontiguousarray(a), np.ascontiguousarray(b))
        return emu().mulmod(self._nsq_arr, np.ascontiguousarray(a), np.ascontiguousarray(b))

    def add_plain(self, c, m):
        if c.shape[0] == 0:
            return c.copy()
        if self._wide():
            return emu().mulmod_n2_split(self._n_arr, np.ascontiguousarray(c), np.ascontiguousarray(m), b_plain=True)
        return emu().add_plain(self._n_arr, np.ascontiguousarray(c), np.ascontiguousarray(m))

    def powmod(self, base, exps):
        if base.shape[0] == 0:
            return base.copy()
        return emu().powmod_n2(self._n_arr, np.ascontiguousarray(base), np.ascontiguousarray(exps))

    def multiexp(self, base, exps):
        return self.multiexp_rows(base, None, np.ascontiguousarray(exps)[None], None)

    def multiexp_rows(self, base, base_inv, exps, neg):
        """(rows, ct_limbs): row r = prod_i b_i^exps[r][i].  k_multiexp_split through the emulator (chunks of 3: a ragged
        last chunk; row blocks of 2), then the pairwise product tree over the chunk index that the library runs with
        k_mulmod; without a split geometry the per-row powmod + tree (as Engine.raw_matvec does on such keys)."""
        rows = exps.shape[0]
        one = np.zeros((rows, self.ct_limbs), np.uint32)
        one[:, 0] = 1
        if base.shape[0] == 0:
            return one
        base, exps = np.ascontiguousarray(base), np.ascontiguousarray(exps)
        parts = emu().multiexp_n2(self._n_arr, base, exps, 3, base_inv=base_inv, neg=neg, row_block=2)
        if parts is None:
            parts = np.zeros((base.shape[0], rows, self.ct_limbs), np.uint32)
            for r in range(rows):
                b = base if neg is None else np.where(np.asarray(neg[r], bool)[:, None], base_inv, base)
                parts[:, r] = emu().powmod_n2(self._n_arr, np.ascontiguousarray(b), np.ascontiguousarray(exps[r]))
        while parts.shape[0] > 1:
            half = parts.shape[0] // 2
            a = np.ascontiguousarray(parts[:half]).reshape(half * rows, -1)
            b = np.ascontiguousarray(parts[half:2 * half]).reshape(half * rows, -1)
            merged = emu().mulmod(self._nsq_arr, a, b).reshape(half, rows, -1)
            parts = np.concatenate([merged, parts[2 * half:]])
        return parts[0]

    def multiexp_csr(self, base, base_inv, row_ptr, cols, exps, neg, order, rows):
        base, exps = np.ascontiguousarray(base), np.ascontiguousarray(exps)
        out = emu().multiexp_csr(self._n_arr, base, exps, base_inv=base_inv, neg=neg, row_ptr=row_ptr, cols=cols, order=order,
                                 rows=rows)
        if out is None:
            raise ValueError("the table form needs the split-modulus engine")
        return out

    # decimal wire format: csrc/radix_conv.h through the emulator library (plain arrays instead of the LDS tile)
    @staticmethod
    def decimal_width(words):
        return emu().L.emu_decimal_width(int(words))

    def to_decimal(self, limbs):
        return emu().to_decimal(limbs)

    def from_decimal(self, digits, words):
        return emu().from_decimal(digits, words)

    def invert(self, a):
        # the product runs a mulmod product tree on the GPU plus one scalar inversion; the emulator
        # backend takes the scalar inverses directly (the tree itself is covered by the GPU tests)
        nsq = self.n * self.n
        out = []
        for i, v in enumerate(self._native.limbs_to_ints(a)):
            try:
                out.append(pow(v, -1, nsq))
            except ValueError:
                err = ZeroDivisionError("invert() no inverse exists")
                err.bad_index = i
                raise err
        return self._native.ints_to_limbs(out, self.ct_limbs)


def _emu_miller_rabin(n, base, device=0):
    return emu().miller_rabin(n, base)


def install(monkeypatch=None):
    from phe import _native
    if monkeypatch is not None:
        monkeypatch.setattr(_native, "Context", EmuContext)
        monkeypatch.setattr(_native, "miller_rabin", _emu_miller_rabin)
    else:
        _native.Context = EmuContext
        _native.miller_rabin = _emu_miller_rabin
