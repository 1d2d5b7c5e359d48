"""ctypes binding of tests/emu/libphe_emu.so (TEST INFRASTRUCTURE: the device algorithm headers compiled
for the CPU on a fiber-based wavefront emulator)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_DIR = os.path.join(HERE, "emu")
LIB = os.path.join(EMU_DIR, "libphe_emu.so")


def P(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class Emu:
    def __init__(self):
        subprocess.check_call(["make", "-C", EMU_DIR, "-s"])
        self.L = ctypes.CDLL(LIB)
        self.L.emu_last_error.restype = ctypes.c_char_p

    def _ck(self, rc):
        if rc != 0:
            raise RuntimeError(self.L.emu_last_error().decode())

    def set_group(self, g):
        self.L.emu_set_group(int(g))

    def set_wave_pairs(self, on):
        """True: on the whole-wave geometry (set_group(64)) encrypt / decrypt run every number on a PAIR of waves"""
        self.L.emu_set_wave_pairs(1 if on else 0)

    def set_late(self, on):
        """True: on the rungs of 16 lanes / the whole wave (set_group(16 | 64)) encrypt, obfuscate and the decrypt halves run on
        the late sweeps (split_core.h modexp_split_late_body: the library's choice for small batches)"""
        self.L.emu_set_late(1 if on else 0)

    def set_wave_tail(self, on):
        """True: the CRT tail of decrypt runs one ciphertext per wavefront (the library's choice for small batches)"""
        self.L.emu_set_wave_tail(1 if on else 0)

    def set_unit(self, on):
        """True (default): r^n through the scaled modulus where the key offers it, as the library's large-batch path does"""
        self.L.emu_set_unit(1 if on else 0)

    def set_engine(self, split):
        """True: split-modulus kernels where a geometry exists (the product default); False: full-width only"""
        self.L.emu_set_engine(1 if split else 0)

    def split_geometry(self, n):
        gl = (ctypes.c_int * 2)()
        self._ck(self.L.emu_split_geometry(P(n), len(n), gl))
        return gl[0], gl[1]

    def private_split_geometry(self, p, q, hp, hq, pinv, n_limbs):
        """(G, L) of the CRT halves' kernels for this key (rung 0 of the private side)"""
        gl = (ctypes.c_int * 2)()
        self._ck(self.L.emu_private_split_geometry(P(p), P(q), P(hp), P(hq), P(pinv), p.shape[0], n_limbs, gl))
        return gl[0], gl[1]

    def montmul(self, G, L, a, b, n, n0inv):
        """a, b: (64/G, G*L) arrays of 29-bit limbs; n: (G*L,) limbs."""
        out = np.zeros((64 // G, G * L), np.uint32)
        self._ck(self.L.emu_montmul(G, L, P(a), P(b), P(n), ctypes.c_uint32(n0inv), P(out)))
        return out

    def encrypt(self, n, m, r):
        s1 = n.shape[0]
        out = np.zeros((m.shape[0], 2 * s1), np.uint32)
        self._ck(self.L.emu_encrypt(P(n), s1, P(m), P(r), None, P(out), ctypes.c_uint64(m.shape[0])))
        return out

    def obfuscate(self, n, c_in, r):
        s1 = n.shape[0]
        out = np.zeros((c_in.shape[0], 2 * s1), np.uint32)
        self._ck(self.L.emu_encrypt(P(n), s1, None, P(r), P(c_in), P(out), ctypes.c_uint64(c_in.shape[0])))
        return out

    def encrypt_owner(self, n, p, q, hp, hq, pinv, m, r):
        """raw_encrypt by the key owner (CRT halves + lift + plaintext factor); None where the library would not offer it"""
        out = np.zeros((m.shape[0], 2 * n.shape[0]), np.uint32)
        rc = self.L.emu_encrypt_owner(P(n), P(p), P(q), P(hp), P(hq), P(pinv), p.shape[0], n.shape[0], P(m), P(r), P(out),
                                      ctypes.c_uint64(m.shape[0]))
        if rc == 2:
            return None
        self._ck(rc)
        return out

    def decrypt(self, p, q, hp, hq, pinv, n_limbs, c):
        out = np.zeros((c.shape[0], n_limbs), np.uint32)
        self._ck(self.L.emu_decrypt(P(p), P(q), P(hp), P(hq), P(pinv), p.shape[0], n_limbs, P(c), P(out),
                                    ctypes.c_uint64(c.shape[0])))
        return out

    def mulmod(self, N, a, b):
        out = np.zeros_like(a)
        self._ck(self.L.emu_mulmod(P(N), N.shape[0], P(a), P(b), P(out), ctypes.c_uint64(a.shape[0])))
        return out

    def mulmod_table(self, N, a, b, tiles=False, blocks=2, waves=16):
        """a*b mod N by the table kernel's body (csrc/mul_table.h: one plain product + one fold against the key's table); None
        where the library would not offer it (the table does not fit a CU's LDS).  tiles: by csrc/mul_tile.h instead (tiles of
        64 products per workgroup of 8 waves, the fold on lane = element), `blocks` emulated workgroups"""
        out = np.zeros_like(a)
        self.L.emu_set_tile_mul((2 if waves == 8 else 1) if tiles else 0, blocks)   # (waves=8: the 512-thread workgroup shape, S = 8 L)
        try:
            rc = self.L.emu_mulmod_table(P(N), N.shape[0], P(a), P(b), P(out), ctypes.c_uint64(a.shape[0]))
        finally:
            self.L.emu_set_tile_mul(0, 0)
        if rc == 2:
            return None
        self._ck(rc)
        return out

    def add_plain(self, n, c, m):
        out = np.zeros_like(c)
        self._ck(self.L.emu_add_plain(P(n), n.shape[0], P(c), P(m), P(out), ctypes.c_uint64(c.shape[0])))
        return out

    def mulmod_n2_split(self, n, a, b, b_plain=False):
        """a*b mod n^2 (or a*(1 + n*m) for plaintext rows b) on the pair form — the product kernel of wide keys; None
        without a split geometry"""
        out = np.zeros_like(a)
        rc = self.L.emu_mulmod_n2_split(P(n), n.shape[0], P(a), P(b), 1 if b_plain else 0, P(out), ctypes.c_uint64(a.shape[0]))
        if rc == 2:
            return None
        self._ck(rc)
        return out

    def powmod_var(self, N, base, exps):
        out = np.zeros_like(base)
        self._ck(self.L.emu_powmod_var(P(N), N.shape[0], P(base), P(exps), exps.shape[1], P(out),
                                       ctypes.c_uint64(base.shape[0])))
        return out

    def split_pair_op(self, G, L, n, op, X, Y=None, out_limbs=0):
        """op 0: X*Y, 1: X^2 (pairs as (64/G, 2*G*L) arrays of 29-bit limbs); 2: canonical value(X) as 32-bit words"""
        groups = 64 // G
        out = np.zeros((groups, out_limbs if op == 2 else 2 * G * L), dtype=np.uint32)
        Yp = P(Y) if Y is not None else None
        self._ck(self.L.emu_split_pair_op(G, L, P(n), len(n), op, P(X), Yp, P(out), out_limbs))
        return out

    def pair_words(self, n):
        w = ctypes.c_int(0)
        rc = self.L.emu_pair_op(P(n), n.shape[0], 0, 0, None, None, 0, None, ctypes.c_uint64(0), ctypes.byref(w))
        return 0 if rc == 2 else w.value

    def pair_op(self, n, op, a, b=None, b_is_row=False, group=0):
        """the pair-form entry points: op 0 words -> pair, 1 pair -> words (b: plaintext rows or None), 2 pair * pair.
        group > 0: on the wider rung of that group width (None if it does not share the rows' limb count)."""
        w = ctypes.c_int(0)
        rc = self.L.emu_pair_op(P(n), n.shape[0], 0, 0, None, None, 0, None, ctypes.c_uint64(0), ctypes.byref(w))
        if rc == 2:
            return None
        self._ck(rc)
        cols = 2 * n.shape[0] if op == 1 else w.value
        a = np.ascontiguousarray(a, np.uint32)
        out = np.zeros((a.shape[0], cols), np.uint32)
        if b is not None:
            b = np.ascontiguousarray(b, np.uint32)
        rc = self.L.emu_pair_op(P(n), n.shape[0], op, group, P(a), P(b) if b is not None else None, 1 if b_is_row else 0,
                                P(out), ctypes.c_uint64(a.shape[0]), None)
        if rc == 2:
            return None
        self._ck(rc)
        return out

    def pair_powmod(self, n, a, exps, group=0):
        """a[i]^exps[i] on pair-form rows, pair form out (phe_hip_pair_powmod_dev); None where pair_op gives None"""
        a = np.ascontiguousarray(a, np.uint32)
        exps = np.ascontiguousarray(exps, np.uint32)
        out = np.zeros_like(a)
        rc = self.L.emu_pair_powmod(P(n), n.shape[0], group, P(a), P(exps), exps.shape[1], P(out), ctypes.c_uint64(a.shape[0]))
        if rc == 2:
            return None
        self._ck(rc)
        return out

    def powmod_n2(self, n, base, exps):
        """base^exp mod n^2 the way phe_hip_powmod runs it (split-modulus kernel when the engine is on)"""
        out = np.zeros_like(base)
        self._ck(self.L.emu_powmod_n2(P(n), n.shape[0], P(base), P(exps), exps.shape[1], P(out),
                                      ctypes.c_uint64(base.shape[0])))
        return out

    def multiexp_n2(self, n, base, exps, chunk, base_inv=None, neg=None, row_block=1, pair_in=False):
        """per-chunk products the way k_multiexp_split forms them -> (n_chunks, rows, ct_limbs); exps: (batch, exp_limbs)
        for one row or (rows, batch, exp_limbs); neg: (rows, batch) bytes selecting base_inv.  None without a split geometry.
        pair_in: `base` holds rows in the pair form (pair_op(..., 0, ...)): no conversion in."""
        B = base.shape[0]
        exps = np.ascontiguousarray(exps.reshape((-1, B, exps.shape[-1])))
        rows = exps.shape[0]
        n_chunks = -(-B // chunk)
        out = np.zeros((n_chunks, rows, 2 * n.shape[0]), dtype=np.uint32)
        self.L.emu_set_multiexp_pair_in(1 if pair_in else 0)
        inv_p = P(np.ascontiguousarray(base_inv)) if base_inv is not None else None
        neg_arr = np.ascontiguousarray(neg, dtype=np.uint8).reshape(rows, B) if neg is not None else None
        rc = self.L.emu_multiexp_n2(P(n), n.shape[0], P(base), inv_p, P(exps), P(neg_arr) if neg_arr is not None else None,
                                    exps.shape[-1], chunk, row_block, P(out), ctypes.c_uint64(rows), ctypes.c_uint64(B))
        self.L.emu_set_multiexp_pair_in(0)
        if rc == 2:
            return None
        self._ck(rc)
        return out

    def multiexp_csr(self, n, base, exps, base_inv=None, neg=None, row_ptr=None, cols=None, order=None, rows=None):
        """k_multiexp_tables + k_multiexp_lookup: (rows, ct_limbs).  exps: (entries, exp_limbs); dense when row_ptr is
        None (entries = rows * batch).  None without a split geometry."""
        base = np.ascontiguousarray(base, dtype=np.uint32)
        exps = np.ascontiguousarray(exps, dtype=np.uint32)
        B = base.shape[0]
        if row_ptr is not None:
            row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64)
            rows = len(row_ptr) - 1
        else:
            rows = rows or exps.shape[0] // B
        out = np.zeros((rows, base.shape[1]), dtype=np.uint32)
        opt = lambda a, dt: P(np.ascontiguousarray(a, dtype=dt)) if a is not None else None
        keep = [opt(base_inv, np.uint32), opt(cols, np.uint32), opt(neg, np.uint8), opt(order, np.uint32)]
        arrs = [np.ascontiguousarray(a, dtype=dt) if a is not None else None
                for a, dt in ((base_inv, np.uint32), (cols, np.uint32), (neg, np.uint8), (order, np.uint32))]
        ptr = [P(a) if a is not None else None for a in arrs]
        rc = self.L.emu_multiexp_csr(P(n), n.shape[0], P(base), ptr[0], ctypes.c_uint64(B),
                                     P(row_ptr) if row_ptr is not None else None, ptr[1], P(exps), ptr[2], exps.shape[1],
                                     ctypes.c_uint64(exps.shape[0]), ptr[3], P(out), ctypes.c_uint64(rows))
        del keep
        if rc == 2:
            return None
        self._ck(rc)
        return out

    def miller_rabin(self, n, base):
        """csrc/primality.h: strong-probable-prime test of n[i] to base[i]; rows of 32-bit words -> bool array"""
        n = np.ascontiguousarray(n, dtype=np.uint32)
        base = np.ascontiguousarray(base, dtype=np.uint32)
        out = np.zeros(n.shape[0], dtype=np.uint8)
        rc = self.L.emu_miller_rabin(P(n), P(base), n.shape[1], P(out), ctypes.c_uint64(n.shape[0]))
        if rc == 2:
            raise ValueError("no 16-lane geometry for %d-bit candidates" % (32 * n.shape[1]))
        self._ck(rc)
        return out.astype(bool)

    def to_decimal(self, limbs, width=None):
        """csrc/radix_conv.h limbs_to_decimal per row -> (rows, width) uint8 ASCII digits, '0'-padded on the left"""
        limbs = np.ascontiguousarray(limbs, dtype=np.uint32)
        width = width or self.L.emu_decimal_width(limbs.shape[1])
        out = np.zeros((limbs.shape[0], width), dtype=np.uint8)
        self.L.emu_to_decimal.restype = ctypes.c_longlong
        bad = self.L.emu_to_decimal(P(limbs), limbs.shape[1], P(out), width, ctypes.c_uint64(limbs.shape[0]))
        if bad >= 0:
            raise ValueError("row %d needs more than %d digits" % (bad, width))
        return out

    def from_decimal(self, digits, words):
        """csrc/radix_conv.h decimal_to_limbs per row of ASCII digits -> (rows, words) uint32"""
        digits = np.ascontiguousarray(digits, dtype=np.uint8)
        out = np.zeros((digits.shape[0], words), dtype=np.uint32)
        bad = ctypes.c_uint64(0)
        st = self.L.emu_from_decimal(P(digits), digits.shape[1], P(out), words, ctypes.c_uint64(digits.shape[0]), ctypes.byref(bad))
        if st:
            err = ValueError("invalid literal: not a decimal digit" if st == 1 else "value does not fit the limb width")
            err.bad_index = bad.value
            raise err
        return out

    def modulus_geometry(self, N):
        GL = (ctypes.c_int * 2)()
        self._ck(self.L.emu_modulus_geometry(P(N), N.shape[0], GL))
        return GL[0], GL[1]

    def public_constants(self, n):
        s1 = n.shape[0]
        res = {}
        GL = (ctypes.c_int * 2)()
        n0 = ctypes.c_uint32(0)
        info = (ctypes.c_int * 6)()
        for which, name in enumerate(["n", "r1", "r2", "r3", "aux"]):
            buf = np.zeros(1024, np.uint32)
            S = self.L.emu_public_constants(P(n), s1, which, P(buf), GL, ctypes.byref(n0), info)
            if S < 0:
                raise RuntimeError(self.L.emu_last_error().decode())
            res[name] = buf[:S].copy()
        res.update(G=GL[0], L=GL[1], S=S, n0inv=n0.value, window=info[0], tbl_entries=info[1], first_idx=info[2],
                   n_ops=info[3], squarings=info[4], multiplies=info[5])
        return res


def to_r29(x, count):
    """Python int -> `count` limbs of 29 bits (numpy uint32)."""
    return np.array([(x >> (29 * j)) & ((1 << 29) - 1) for j in range(count)], dtype=np.uint32)


def from_r29(limbs):
    """limbs (possibly not normalised) -> Python int."""
    return sum(int(v) << (29 * j) for j, v in enumerate(np.asarray(limbs).tolist()))
