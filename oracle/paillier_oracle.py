"""oracle/paillier_oracle.py — CPU oracle, Python side.  TEST INFRASTRUCTURE ONLY.

Two independent restatements of the reference hot path (paths relative to /root/reference):

* ``Py*`` functions: pure-Python-int restatement (CPython ``pow`` is the engine the
  reference itself falls back to, phe/util.py:48, :61, :100-103).  Small cases only.
* ``COracle``: ctypes binding of oracle/libphe_oracle.so (paillier_oracle.c, libgmp —
  the engine gmpy2 wraps).  Batch-capable, threaded; used as checker and CPU baseline.

Both are pinned in tests/test_oracle.py against the reference's known-answer vectors and
against tests/golden/ fixtures generated from the real reference (tests/golden/gen_golden.py).

Who may import this: tests/, __graft_entry__.smoke(), bench.py (checker + cpu_baseline leg) and the measurement scripts
under tools/ (bench_*.py, ref_cpu_baseline.py — the same role as bench.py: the checker beside a timed HIP run, or the CPU
figure printed next to it).  Nothing under python-paillier_amd/ does, and the package has no CPU route at all.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libphe_oracle.so")


# --------------------------------------------------------------------------
# limb helpers (the layout of include/phe_hip.h: little-endian uint32 limbs)
# --------------------------------------------------------------------------
def int_to_limbs(x, limbs):
    return np.frombuffer(int(x).to_bytes(4 * limbs, "little"), dtype=np.uint32).copy()


def ints_to_limbs(xs, limbs):
    buf = b"".join(int(x).to_bytes(4 * limbs, "little") for x in xs)
    return np.frombuffer(buf, dtype=np.uint32).reshape(len(xs), limbs).copy()


def limbs_to_int(row):
    return int.from_bytes(np.ascontiguousarray(row, dtype=np.uint32).tobytes(), "little")


def limbs_to_ints(arr):
    arr = np.ascontiguousarray(arr, dtype=np.uint32)
    nbytes = arr.shape[1] * 4
    raw = arr.tobytes()
    return [int.from_bytes(raw[i * nbytes:(i + 1) * nbytes], "little") for i in range(arr.shape[0])]


# --------------------------------------------------------------------------
# pure-Python restatement
# --------------------------------------------------------------------------
def py_powmod(a, b, c):
    """phe/util.py:38-50"""
    if a == 1:
        return 1
    return pow(a, b, c)


def py_mulmod(a, b, c):
    """phe/util.py:53-64"""
    return a * b % c


def py_invert(a, b):
    """phe/util.py:85-103 (extended Euclid branch :100-103)"""
    r0, r1, s0, s1 = a, b, 1, 0
    while r1 != 0:
        qt = r0 // r1
        r0, r1 = r1, r0 - qt * r1
        s0, s1 = s1, s0 - qt * s1
    if r0 != 1:
        raise ZeroDivisionError("invert() no inverse exists")
    return s0 % b


class PyPublic:
    """phe/paillier.py:86-90"""

    def __init__(self, n):
        self.n = n
        self.g = n + 1
        self.nsquare = n * n
        self.max_int = n // 3 - 1

    def raw_encrypt(self, m, r):
        """phe/paillier.py:102-139 with r_value explicit"""
        n, nsq = self.n, self.nsquare
        if n - self.max_int <= m < n:
            nude = py_invert((n * (n - m) + 1) % nsq, nsq)
        else:
            nude = (n * m + 1) % nsq
        return py_mulmod(nude, py_powmod(r, n, nsq), nsq)

    def obfuscate(self, c, r):
        """phe/paillier.py:603-624 with r explicit"""
        return py_mulmod(c, py_powmod(r, self.n, self.nsquare), self.nsquare)

    def raw_add(self, a, b):
        """phe/paillier.py:705-719"""
        return py_mulmod(a, b, self.nsquare)

    def raw_mul(self, c, s):
        """phe/paillier.py:721-751"""
        if s < 0 or s >= self.n:
            raise ValueError("Scalar out of bounds: %i" % s)
        if self.n - self.max_int <= s:
            return py_powmod(py_invert(c, self.nsquare), self.n - s, self.nsquare)
        return py_powmod(c, s, self.nsquare)


class PyPrivate:
    """phe/paillier.py:217-235"""

    def __init__(self, pub, p, q):
        if p * q != pub.n:
            raise ValueError("given public key does not match the given p and q.")
        if p == q:
            raise ValueError("p and q have to be different")
        self.pub = pub
        self.p, self.q = (q, p) if q < p else (p, q)
        self.psquare = self.p * self.p
        self.qsquare = self.q * self.q
        self.p_inverse = py_invert(self.p, self.q)
        self.hp = self._h(self.p, self.psquare)
        self.hq = self._h(self.q, self.qsquare)

    def _h(self, x, xsq):
        """phe/paillier.py:356-360"""
        return py_invert((py_powmod(self.pub.g, x - 1, xsq) - 1) // x, x)

    def raw_decrypt(self, c):
        """phe/paillier.py:328-354, crt :366-374"""
        mp = py_mulmod((py_powmod(c, self.p - 1, self.psquare) - 1) // self.p, self.hp, self.p)
        mq = py_mulmod((py_powmod(c, self.q - 1, self.qsquare) - 1) // self.q, self.hq, self.q)
        u = py_mulmod(mq - mp, self.p_inverse, self.q)
        return mp + u * self.p


# --------------------------------------------------------------------------
# C oracle (libgmp) binding
# --------------------------------------------------------------------------
def build_c_oracle(force=False):
    """Compile oracle/libphe_oracle.so (gcc + system libgmp). Building the checker is not using it."""
    src = os.path.join(_HERE, "paillier_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")


class COracle:
    def __init__(self):
        if not os.path.exists(_LIB_PATH):
            build_c_oracle()
        L = ctypes.CDLL(_LIB_PATH)
        sz = ctypes.c_size_t
        L.orc_powmod.argtypes = [_u32p, _u32p, _u32p, _u32p, sz]
        L.orc_mulmod.argtypes = [_u32p, _u32p, _u32p, _u32p, sz]
        L.orc_invert.argtypes = [_u32p, _u32p, _u32p, sz]
        L.orc_private_constants.argtypes = [_u32p, sz, _u32p, _u32p, sz, _u32p, _u32p, _u32p, _u32p, _u32p]
        L.orc_encrypt_batch.argtypes = [_u32p, sz, _u32p, _u32p, _u32p, sz, ctypes.c_int]
        L.orc_obfuscate_batch.argtypes = [_u32p, sz, _u32p, _u32p, _u32p, sz, ctypes.c_int]
        L.orc_decrypt_batch.argtypes = [_u32p, sz, _u32p, _u32p, sz, _u32p, _u32p, sz, ctypes.c_int]
        L.orc_add_batch.argtypes = [_u32p, sz, _u32p, _u32p, _u32p, sz, ctypes.c_int]
        L.orc_mul_batch.argtypes = [_u32p, sz, _u32p, _u32p, sz, _u32p, sz, ctypes.c_int,
                                    ctypes.POINTER(ctypes.c_size_t)]
        L.orc_next_prime.argtypes = [_u32p, _u32p, sz]
        L.orc_probab_prime.argtypes = [_u32p, sz, ctypes.c_int]
        L.orc_gmp_version.restype = ctypes.c_char_p
        self.L = L

    @property
    def gmp_version(self):
        return self.L.orc_gmp_version().decode()

    # scalar primitives on Python ints
    def powmod(self, a, b, c):
        limbs = max(1, (max(a, b, c).bit_length() + 31) // 32)
        out = np.zeros(limbs, np.uint32)
        self.L.orc_powmod(int_to_limbs(a, limbs), int_to_limbs(b, limbs), int_to_limbs(c, limbs), out, limbs)
        return limbs_to_int(out)

    def mulmod(self, a, b, c):
        limbs = max(1, (max(a, b, c).bit_length() + 31) // 32)
        out = np.zeros(limbs, np.uint32)
        self.L.orc_mulmod(int_to_limbs(a, limbs), int_to_limbs(b, limbs), int_to_limbs(c, limbs), out, limbs)
        return limbs_to_int(out)

    def invert(self, a, b):
        limbs = max(1, (max(a, b).bit_length() + 31) // 32)
        out = np.zeros(limbs, np.uint32)
        rc = self.L.orc_invert(int_to_limbs(a, limbs), int_to_limbs(b, limbs), out, limbs)
        if rc:
            raise ZeroDivisionError("invert() no inverse exists")
        return limbs_to_int(out)

    def next_prime(self, start):
        """gmpy2.next_prime(start) (phe/util.py:116): the smallest probable prime above `start`"""
        limbs = start.bit_length() // 32 + 2
        out = np.zeros(limbs, np.uint32)
        if self.L.orc_next_prime(int_to_limbs(start, limbs), out, limbs):
            raise OverflowError("next prime does not fit")
        return limbs_to_int(out)

    def is_probable_prime(self, n, reps=25):
        limbs = max(1, (n.bit_length() + 31) // 32)
        return self.L.orc_probab_prime(int_to_limbs(n, limbs), limbs, reps) > 0

    def private_constants(self, n, p, q, n_limbs, pq_limbs):
        outs = [np.zeros(pq_limbs, np.uint32) for _ in range(5)]
        rc = self.L.orc_private_constants(int_to_limbs(n, n_limbs), n_limbs, int_to_limbs(p, pq_limbs),
                                          int_to_limbs(q, pq_limbs), pq_limbs, *outs)
        if rc:
            raise ZeroDivisionError("invert() no inverse exists")
        return tuple(limbs_to_int(o) for o in outs)  # p, q, hp, hq, p_inverse

    # batch ops on limb arrays
    def encrypt(self, n_limbs_arr, m, r, nthreads=1):
        s1 = n_limbs_arr.shape[0]
        B = m.shape[0]
        c = np.zeros((B, 2 * s1), np.uint32)
        rc = self.L.orc_encrypt_batch(n_limbs_arr, s1, np.ascontiguousarray(m), np.ascontiguousarray(r), c, B, nthreads)
        assert rc == 0, rc
        return c

    def obfuscate(self, n_limbs_arr, c_in, r, nthreads=1):
        s1 = n_limbs_arr.shape[0]
        B = c_in.shape[0]
        c = np.zeros((B, 2 * s1), np.uint32)
        rc = self.L.orc_obfuscate_batch(n_limbs_arr, s1, np.ascontiguousarray(c_in), np.ascontiguousarray(r), c, B, nthreads)
        assert rc == 0, rc
        return c

    def decrypt(self, n_limbs_arr, p_limbs, q_limbs, c, nthreads=1):
        s1 = n_limbs_arr.shape[0]
        B = c.shape[0]
        m = np.zeros((B, s1), np.uint32)
        rc = self.L.orc_decrypt_batch(n_limbs_arr, s1, p_limbs, q_limbs, p_limbs.shape[0],
                                      np.ascontiguousarray(c), m, B, nthreads)
        assert rc == 0, rc
        return m

    def add(self, n_limbs_arr, a, b, nthreads=1):
        s1 = n_limbs_arr.shape[0]
        B = a.shape[0]
        out = np.zeros((B, 2 * s1), np.uint32)
        rc = self.L.orc_add_batch(n_limbs_arr, s1, np.ascontiguousarray(a), np.ascontiguousarray(b), out, B, nthreads)
        assert rc == 0, rc
        return out

    def mul(self, n_limbs_arr, c, scalars, nthreads=1):
        s1 = n_limbs_arr.shape[0]
        B = c.shape[0]
        out = np.zeros((B, 2 * s1), np.uint32)
        bad = ctypes.c_size_t(0)
        rc = self.L.orc_mul_batch(n_limbs_arr, s1, np.ascontiguousarray(c), np.ascontiguousarray(scalars),
                                  scalars.shape[1], out, B, nthreads, ctypes.byref(bad))
        if rc == 1:
            raise ValueError("Scalar out of bounds (row %d)" % bad.value)
        if rc == 3:
            raise ZeroDivisionError("invert() no inverse exists (row %d)" % bad.value)
        return out
