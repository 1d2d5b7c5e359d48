"""ctypes binding of the C-ABI in include/phe_hip.h (lib/libphe_hip.so, hand-written HIP for gfx950).

There is deliberately NO CPU fallback here: if the shared library is missing, or a call fails,
an exception is raised.  Status codes map onto the exception types the reference raises
(SURVEY.md 8(b) "Error conventions"): ENOINVERSE -> ZeroDivisionError (phe/util.py:96-102),
EINVAL -> ValueError, EHIP -> RuntimeError.
"""
import ctypes
import functools
import inspect
import os
import threading

import numpy as np

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# PHE_HIP_LIB: another build of the same library (measurement variants of tools/exp/build_variants.sh); default lib/libphe_hip.so
LIB_PATH = os.environ.get("PHE_HIP_LIB") or os.path.join(_PKG_ROOT, "lib", "libphe_hip.so")

OK, EINVAL, EHIP, ENOINVERSE = 0, 1, 2, 3
ABI_VERSION = 6          # include/phe_hip.h PHE_HIP_ABI_VERSION as mirrored below (tests/test_abi_exports.py compares the two)

_lib = None


class NativeLibraryMissing(ImportError):
    pass


def lib():
    """Load lib/libphe_hip.so once; fail loudly if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryMissing(
                "%s not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
                "This package has no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        # the argtypes below mirror include/phe_hip.h at PHE_HIP_ABI_VERSION = ABI_VERSION: a library built from another revision of
        # the header would still link (argument lists have changed under unchanged names) and read shifted arguments
        try:
            L.phe_hip_abi_version.argtypes = []
            have = L.phe_hip_abi_version()
        except AttributeError:
            have = None
        if have != ABI_VERSION:
            raise NativeLibraryMissing("%s speaks ABI version %s, this package mirrors version %d of include/phe_hip.h: rebuild "
                                       "(python __graft_entry__.py build)" % (LIB_PATH, have, ABI_VERSION))
        L.phe_hip_last_error.restype = ctypes.c_char_p
        L.phe_hip_device_count.argtypes = [ctypes.POINTER(ci)]
        L.phe_hip_ctx_create_public.argtypes = [vp, ci, ci, ctypes.POINTER(vp)]
        L.phe_hip_ctx_create_private.argtypes = [vp, ci, vp, vp, vp, vp, vp, ci, ci, ctypes.POINTER(vp)]
        L.phe_hip_ctx_destroy.argtypes = [vp]
        L.phe_hip_ctx_destroy.restype = None
        L.phe_hip_ctx_info.argtypes = [vp] + [ctypes.POINTER(ci)] * 6
        L.phe_hip_ctx_engine.argtypes = [vp] + [ctypes.POINTER(ci)] * 2
        L.phe_hip_ctx_set_blocks_per_cu.argtypes = [vp, ci]
        L.phe_hip_encrypt.argtypes = [vp, vp, vp, vp, sz]
        L.phe_hip_encrypt_owner.argtypes = [vp, vp, vp, vp, sz]
        L.phe_hip_encrypt_owner_dev.argtypes = [vp, vp, vp, vp, sz, vp]
        L.phe_hip_ctx_owner_encrypt.argtypes = [vp, ctypes.POINTER(ci)]
        L.phe_hip_obfuscate.argtypes = [vp, vp, vp, vp, sz]
        L.phe_hip_decrypt.argtypes = [vp, vp, vp, sz]
        L.phe_hip_mulmod.argtypes = [vp, vp, vp, vp, sz]
        L.phe_hip_powmod.argtypes = [vp, vp, vp, ci, vp, sz]
        L.phe_hip_add_plain.argtypes = [vp, vp, vp, vp, sz]
        L.phe_hip_multiexp.argtypes = [vp, vp, vp, ci, vp, sz]
        L.phe_hip_decimal_width.argtypes = [ci]
        L.phe_hip_to_decimal.argtypes = [vp, vp, ci, vp, ci, sz]
        L.phe_hip_from_decimal.argtypes = [vp, vp, ci, vp, ci, sz, ctypes.POINTER(sz)]
        L.phe_hip_to_decimal_dev.argtypes = [vp, vp, ci, vp, ci, sz, vp]
        L.phe_hip_from_decimal_dev.argtypes = [vp, vp, ci, vp, ci, sz, ctypes.POINTER(sz), vp]
        L.phe_hip_multiexp_dev.argtypes = [vp, vp, vp, ci, ci, vp, sz, vp]
        L.phe_hip_multiexp_rows_dev.argtypes = [vp, vp, vp, vp, vp, ci, ci, vp, sz, sz, vp]
        L.phe_hip_multiexp_csr_dev.argtypes = [vp, vp, vp, sz, vp, vp, vp, vp, ci, ci, vp, vp, sz, vp]
        L.phe_hip_add_plain_dev.argtypes = [vp, vp, vp, vp, sz, vp]
        L.phe_hip_invert.argtypes = [vp, vp, vp, sz, ctypes.POINTER(sz)]
        L.phe_hip_encrypt_dev.argtypes = [vp, vp, vp, vp, sz, vp]
        L.phe_hip_obfuscate_dev.argtypes = [vp, vp, vp, vp, sz, vp]
        L.phe_hip_decrypt_dev.argtypes = [vp, vp, vp, sz, vp]
        L.phe_hip_mulmod_dev.argtypes = [vp, vp, vp, vp, sz, vp]
        L.phe_hip_powmod_dev.argtypes = [vp, vp, vp, ci, ci, vp, sz, vp]
        L.phe_hip_montmul_dev.argtypes = [vp, vp, vp, ci, vp, sz, vp]
        L.phe_hip_mont_radix_bits.argtypes = [vp, ctypes.POINTER(ci)]
        L.phe_hip_malloc.argtypes = [vp, sz, ctypes.POINTER(vp)]
        L.phe_hip_free.argtypes = [vp, vp]
        L.phe_hip_memcpy_h2d.argtypes = [vp, vp, vp, sz]
        L.phe_hip_memcpy_d2h.argtypes = [vp, vp, vp, sz]
        L.phe_hip_stream_sync.argtypes = [vp, vp]
        L.phe_hip_stream_create.argtypes = [vp, ctypes.POINTER(vp)]
        L.phe_hip_miller_rabin.argtypes = [ci, vp, vp, ci, vp, sz]
        L.phe_hip_stream_destroy.argtypes = [vp, vp]
        L.phe_hip_memcpy_d2d.argtypes = [vp, vp, vp, sz, vp]
        L.phe_hip_invert_dev.argtypes = [vp, vp, vp, sz, ctypes.POINTER(sz), vp]
        L.phe_hip_select_rows_dev.argtypes = [vp, vp, vp, vp, vp, ci, sz, vp]
        L.phe_hip_gather_rows_dev.argtypes = [vp, vp, sz, vp, vp, ci, sz, vp]
        L.phe_hip_scatter_rows_dev.argtypes = [vp, vp, vp, vp, sz, ci, sz, vp]
        L.phe_hip_selftest_prims.argtypes = [ci, vp]
        L.phe_hip_comm_unique_id.argtypes = [vp]
        L.phe_hip_comm_create.argtypes = [vp, vp, ci, ci, ctypes.POINTER(vp)]
        L.phe_hip_allgather_dev.argtypes = [vp, vp, vp, sz, ci, vp]
        L.phe_hip_comm_destroy.argtypes = [vp]
        L.phe_hip_comm_destroy.restype = None
        L.phe_hip_ctx_ladder.argtypes = [vp, vp, vp, ci, ctypes.POINTER(ci), ctypes.POINTER(ci)]
        L.phe_hip_ctx_set_group.argtypes = [vp, ci]
        L.phe_hip_ctx_load_ladder.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(ci)]
        L.phe_hip_ctx_last_launch.argtypes = [vp] + [ctypes.POINTER(ci)] * 3
        L.phe_hip_ctx_release_scratch.argtypes = [vp]
        L.phe_hip_pair_words.argtypes = [vp, ctypes.POINTER(ci)]
        L.phe_hip_to_pair_dev.argtypes = [vp, vp, vp, sz, vp]
        L.phe_hip_pair_mul_dev.argtypes = [vp, vp, vp, ci, vp, sz, vp]
        L.phe_hip_from_pair_dev.argtypes = [vp, vp, vp, vp, sz, vp]
        L.phe_hip_pair_reduce_dev.argtypes = [vp, vp, sz, vp, vp]
        L.phe_hip_pair_powmod_dev.argtypes = [vp, vp, vp, ci, ci, vp, sz, vp]
        L.phe_hip_pair_multiexp_rows_dev.argtypes = [vp, vp, vp, ci, ci, vp, sz, sz, vp]
        _lib = L
    return _lib


EXPORTED_SYMBOLS = [
    "phe_hip_last_error", "phe_hip_abi_version", "phe_hip_device_count", "phe_hip_ctx_create_public", "phe_hip_ctx_create_private",
    "phe_hip_ctx_destroy", "phe_hip_ctx_info", "phe_hip_ctx_set_blocks_per_cu", "phe_hip_encrypt",
    "phe_hip_obfuscate", "phe_hip_decrypt", "phe_hip_mulmod", "phe_hip_powmod", "phe_hip_invert",
    "phe_hip_encrypt_dev", "phe_hip_obfuscate_dev", "phe_hip_decrypt_dev", "phe_hip_mulmod_dev",
    "phe_hip_powmod_dev", "phe_hip_malloc", "phe_hip_free", "phe_hip_memcpy_h2d", "phe_hip_memcpy_d2h",
    "phe_hip_stream_sync", "phe_hip_selftest_prims", "phe_hip_memcpy_d2d", "phe_hip_invert_dev",
    "phe_hip_select_rows_dev", "phe_hip_gather_rows_dev", "phe_hip_scatter_rows_dev", "phe_hip_add_plain", "phe_hip_add_plain_dev", "phe_hip_ctx_engine",
    "phe_hip_multiexp", "phe_hip_multiexp_dev", "phe_hip_multiexp_rows_dev", "phe_hip_multiexp_csr_dev", "phe_hip_decimal_width", "phe_hip_to_decimal",
    "phe_hip_from_decimal", "phe_hip_to_decimal_dev", "phe_hip_from_decimal_dev", "phe_hip_stream_create",
    "phe_hip_stream_destroy", "phe_hip_miller_rabin", "phe_hip_montmul_dev", "phe_hip_mont_radix_bits",
    "phe_hip_comm_unique_id", "phe_hip_comm_create", "phe_hip_allgather_dev", "phe_hip_comm_destroy",
    "phe_hip_encrypt_owner", "phe_hip_encrypt_owner_dev", "phe_hip_ctx_owner_encrypt",
    "phe_hip_ctx_ladder", "phe_hip_ctx_set_group", "phe_hip_ctx_load_ladder", "phe_hip_ctx_last_launch", "phe_hip_ctx_release_scratch",
    "phe_hip_pair_words", "phe_hip_to_pair_dev", "phe_hip_pair_mul_dev", "phe_hip_from_pair_dev", "phe_hip_pair_reduce_dev",
    "phe_hip_pair_powmod_dev", "phe_hip_pair_multiexp_rows_dev",
]


LADDER_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ladder_gfx950.txt")
_ladder_text = None


def measured_ladder_text():
    """the committed calibration of the geometry ladder (tools/bench_sweep.py --calibrate on an MI355X), or None.
    PHE_HIP_LADDER_FILE names another table; PHE_HIP_NO_MEASURED_LADDER=1 keeps the library's estimate."""
    global _ladder_text
    if _ladder_text is None:
        try:
            with open(os.environ.get("PHE_HIP_LADDER_FILE", LADDER_FILE)) as f:
                _ladder_text = f.read()
        except OSError:
            _ladder_text = ""
    return _ladder_text or None


def _raise(rc, bad_index=None):
    msg = lib().phe_hip_last_error().decode(errors="replace")
    if rc == ENOINVERSE:
        err = ZeroDivisionError(msg)
        err.bad_index = bad_index
        raise err
    if rc == EINVAL:
        raise ValueError(msg)
    raise RuntimeError("HIP error: " + msg)


def _check(rc):
    if rc != OK:
        _raise(rc)


def max_exp_bits(exps):
    """largest bit length among exponent rows (..., exp_limbs) of little-endian uint32 words (at least 1)"""
    flat = np.asarray(exps).reshape(-1, exps.shape[-1])
    for k in range(flat.shape[1] - 1, -1, -1):
        top = int(flat[:, k].max()) if flat.shape[0] else 0
        if top:
            return 32 * k + top.bit_length()
    return 1


def int_to_limbs(x, limbs):
    return np.frombuffer(int(x).to_bytes(4 * limbs, "little"), dtype=np.uint32).copy()


def ints_to_limbs(xs, limbs):
    nbytes = 4 * limbs
    buf = bytearray(nbytes * len(xs))
    for i, x in enumerate(xs):
        buf[i * nbytes:(i + 1) * nbytes] = int(x).to_bytes(nbytes, "little")
    return np.frombuffer(bytes(buf), dtype=np.uint32).reshape(len(xs), limbs).copy()


def limbs_to_ints(arr):
    arr = np.ascontiguousarray(arr, dtype=np.uint32)
    if arr.ndim == 1:
        arr = arr.reshape(1, -1)
    nbytes = arr.shape[1] * 4
    raw = arr.tobytes()
    return [int.from_bytes(raw[i * nbytes:(i + 1) * nbytes], "little") for i in range(arr.shape[0])]


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _rows(a, limbs, name):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    if a.ndim != 2 or a.shape[1] != limbs:
        raise ValueError("%s must have shape (batch, %d) uint32, got %r" % (name, limbs, a.shape))
    return a


def device_count():
    n = ctypes.c_int(0)
    _check(lib().phe_hip_device_count(ctypes.byref(n)))
    return n.value


def limbs_for_bits(bits):
    return max(1, (bits + 31) // 32)


def serialised(cls):
    """Class decorator: every public method runs under the instance's re-entrant lock `self._lock`.  ctypes drops the
    GIL inside each native call and a native context (its window tables, launch streams, the size-keyed block pool)
    is not thread-safe, while the reference's functions are stateless and callable from any thread — so calls on one
    key are serialised here instead of being left to the caller.

    Generator methods are left alone: a lock held across `yield` would block every other thread on the key while the
    consumer works on a chunk, and would be orphaned if the generator were finalised on another thread.  They take
    `self._lock` around each of their own steps instead (every native call they make is locked by the Context anyway, and
    the library keeps one stream order per context, so work another thread slips in between two steps cannot race on the
    context's scratch)."""
    for name, fn in list(vars(cls).items()):
        if name.startswith("_") or not inspect.isfunction(fn) or inspect.isgeneratorfunction(fn):
            continue

        def wrap(fn):
            @functools.wraps(fn)
            def locked(self, *a, **kw):
                with self._lock:
                    return fn(self, *a, **kw)
            return locked
        setattr(cls, name, wrap(fn))
    return cls


@serialised
class Context:
    """One key on one device.  `n` (and optionally p, q, hp, hq, p_inverse) are Python ints."""

    def __init__(self, n, p=None, q=None, hp=None, hq=None, p_inverse=None, device=0, n_limbs=None):
        self._lock = threading.RLock()
        L = lib()
        self.n = int(n)
        self.n_limbs = n_limbs or limbs_for_bits(self.n.bit_length())
        self.ct_limbs = 2 * self.n_limbs
        self.device = device
        self._h = ctypes.c_void_p(None)
        n_arr = int_to_limbs(self.n, self.n_limbs)
        if p is None:
            _check(L.phe_hip_ctx_create_public(_ptr(n_arr), self.n_limbs, device, ctypes.byref(self._h)))
            self.has_private = False
        else:
            pq = limbs_for_bits(max(int(p).bit_length(), int(q).bit_length()))
            arrs = [int_to_limbs(v, pq) for v in (p, q, hp, hq, p_inverse)]
            _check(L.phe_hip_ctx_create_private(_ptr(n_arr), self.n_limbs, *[_ptr(a) for a in arrs], pq, device,
                                                ctypes.byref(self._h)))
            self.has_private = True
        self.measured_ladder_lines = 0
        if not os.environ.get("PHE_HIP_NO_MEASURED_LADDER"):
            # an optimisation, never a reason for a key to fail: a malformed table (PHE_HIP_LADDER_FILE) leaves the estimate in
            # place; a table measured on another chip (its header names arch and CU count) is not taken by the library
            try:
                self.load_ladder(measured_ladder_text())
            except (ValueError, RuntimeError) as e:
                self.measured_ladder_lines = 0
                self.ladder_error = str(e)

    def load_ladder(self, text):
        """Give the context a measured ladder (include/phe_hip.h phe_hip_ctx_load_ladder: lines "key_bits family G rows ns"; None
        forgets it: the rung of a call is then picked from the estimate).  Returns the lines kept (those of this key width)."""
        kept = ctypes.c_int(0)
        _check(lib().phe_hip_ctx_load_ladder(self._h, text.encode() if text else None, ctypes.byref(kept)))
        self.measured_ladder_lines = kept.value
        return kept.value

    def close(self):
        if self._h and self._h.value:
            self.trim_pool()
            lib().phe_hip_ctx_destroy(self._h)
            self._h = ctypes.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        vals = [ctypes.c_int(0) for _ in range(6)]
        _check(lib().phe_hip_ctx_info(self._h, *[ctypes.byref(v) for v in vals]))
        keys = ["n_limbs", "ct_limbs", "lane_limbs_pub", "lane_limbs_priv", "rows_in_flight", "has_private"]
        out = dict(zip(keys, [v.value for v in vals]))
        eng = [ctypes.c_int(0), ctypes.c_int(0)]
        _check(lib().phe_hip_ctx_engine(self._h, ctypes.byref(eng[0]), ctypes.byref(eng[1])))
        out["engine_pub"] = "split" if eng[0].value else "full"
        out["engine_priv"] = ("split" if eng[1].value else "full") if out["has_private"] else None
        return out

    def set_blocks_per_cu(self, k):
        _check(lib().phe_hip_ctx_set_blocks_per_cu(self._h, int(k)))

    def ladder(self):
        """the geometry ladder: ([G*100 + L, ...] of the n-side kernels, the same for the p/q side), narrowest groups first"""
        cap = 8
        pub, priv = (ctypes.c_int * cap)(), (ctypes.c_int * cap)()
        n_pub, n_priv = ctypes.c_int(0), ctypes.c_int(0)
        _check(lib().phe_hip_ctx_ladder(self._h, pub, priv, cap, ctypes.byref(n_pub), ctypes.byref(n_priv)))
        return list(pub[:min(cap, n_pub.value)]), list(priv[:min(cap, n_priv.value)])

    def set_group(self, group):
        """0: the rung is picked from the batch size; G: always the rung of G-lane limb groups (tests, measurements)"""
        _check(lib().phe_hip_ctx_set_group(self._h, int(group)))

    PATH_UNIT, PATH_OWNER, PATH_SIDE_BY_SIDE, PATH_PIPELINED, PATH_FUSED_OBFUSCATE, PATH_WAVE_PAIRS, PATH_WAVE_TAIL = 1, 2, 4, 8, 16, 32, 64
    PATH_LATE = 128
    PATH_TABLE_MUL = 256
    PATH_TILE_MUL = 512

    def last_launch(self):
        """what the last encrypt / obfuscate / decrypt / pair call took: {"path": PATH_* bits, "geom_pub", "geom_priv"}"""
        v = [ctypes.c_int(0) for _ in range(3)]
        _check(lib().phe_hip_ctx_last_launch(self._h, *[ctypes.byref(x) for x in v]))
        return {"path": v[0].value, "geom_pub": v[1].value, "geom_priv": v[2].value}

    def release_scratch(self):
        _check(lib().phe_hip_ctx_release_scratch(self._h))

    # ---- resident rows in the pair form (include/phe_hip.h "pair form") ----
    def pair_words(self):
        """words of a pair-form row, or 0 when the context has no split-modulus engine"""
        w = ctypes.c_int(0)
        rc = lib().phe_hip_pair_words(self._h, ctypes.byref(w))
        return w.value if rc == OK else 0

    def to_pair_dev(self, c_ptr, pair_ptr, batch, stream=0):
        _check(lib().phe_hip_to_pair_dev(self._h, c_ptr, pair_ptr, batch, stream))

    def pair_mul_dev(self, a_ptr, b_ptr, b_is_row, out_ptr, batch, stream=0):
        _check(lib().phe_hip_pair_mul_dev(self._h, a_ptr, b_ptr, 1 if b_is_row else 0, out_ptr, batch, stream))

    def from_pair_dev(self, pair_ptr, m_ptr, c_ptr, batch, stream=0):
        _check(lib().phe_hip_from_pair_dev(self._h, pair_ptr, m_ptr, c_ptr, batch, stream))

    def pair_powmod_dev(self, a_ptr, e_ptr, exp_limbs, max_exp_bits, out_ptr, batch, stream=0):
        _check(lib().phe_hip_pair_powmod_dev(self._h, a_ptr, e_ptr, exp_limbs, max_exp_bits, out_ptr, batch, stream))

    def pair_multiexp_rows_dev(self, pair_ptr, exp_ptr, exp_limbs, max_exp_bits, out_ptr, batch, rows, stream=0):
        """rows dot products over a resident vector in the pair form (non-negative scalars): canonical residues out"""
        _check(lib().phe_hip_pair_multiexp_rows_dev(self._h, pair_ptr, exp_ptr, exp_limbs, max_exp_bits, out_ptr, batch, rows, stream))

    def pair_reduce_dev(self, pair_ptr, batch, out_ptr, stream=0):
        _check(lib().phe_hip_pair_reduce_dev(self._h, pair_ptr, batch, out_ptr, stream))

    # ---- host-array entry points (numpy uint32, row-major (batch, limbs)) ----
    def encrypt(self, m, r):
        m = _rows(m, self.n_limbs, "m")
        r = _rows(r, self.n_limbs, "r")
        if m.shape[0] != r.shape[0]:
            raise ValueError("m and r batch sizes differ")
        c = np.empty((m.shape[0], self.ct_limbs), np.uint32)
        _check(lib().phe_hip_encrypt(self._h, _ptr(m), _ptr(r), _ptr(c), m.shape[0]))
        return c

    def owner_encrypt_offered(self):
        """True on a private-key context whose key width has the geometries of the CRT form of raw_encrypt"""
        flag = ctypes.c_int(0)
        _check(lib().phe_hip_ctx_owner_encrypt(self._h, ctypes.byref(flag)))
        return bool(flag.value)

    def encrypt_owner(self, m, r):
        """raw_encrypt by the key owner (r^n from its CRT halves): the same bits as encrypt()"""
        m = _rows(m, self.n_limbs, "m")
        r = _rows(r, self.n_limbs, "r")
        if m.shape[0] != r.shape[0]:
            raise ValueError("m and r batch sizes differ")
        c = np.empty((m.shape[0], self.ct_limbs), np.uint32)
        _check(lib().phe_hip_encrypt_owner(self._h, _ptr(m), _ptr(r), _ptr(c), m.shape[0]))
        return c

    def encrypt_owner_dev(self, m_ptr, r_ptr, c_ptr, batch, stream=0):
        _check(lib().phe_hip_encrypt_owner_dev(self._h, m_ptr, r_ptr, c_ptr, batch, stream))

    def obfuscate(self, c_in, r):
        c_in = _rows(c_in, self.ct_limbs, "c_in")
        r = _rows(r, self.n_limbs, "r")
        if c_in.shape[0] != r.shape[0]:
            raise ValueError("c_in and r batch sizes differ")
        c = np.empty_like(c_in)
        _check(lib().phe_hip_obfuscate(self._h, _ptr(c_in), _ptr(r), _ptr(c), c_in.shape[0]))
        return c

    def decrypt(self, c):
        c = _rows(c, self.ct_limbs, "c")
        m = np.empty((c.shape[0], self.n_limbs), np.uint32)
        _check(lib().phe_hip_decrypt(self._h, _ptr(c), _ptr(m), c.shape[0]))
        return m

    def mulmod(self, a, b):
        a = _rows(a, self.ct_limbs, "a")
        b = _rows(b, self.ct_limbs, "b")
        if a.shape[0] != b.shape[0]:
            raise ValueError("a and b batch sizes differ")
        out = np.empty_like(a)
        _check(lib().phe_hip_mulmod(self._h, _ptr(a), _ptr(b), _ptr(out), a.shape[0]))
        return out

    def add_plain(self, c, m):
        c = _rows(c, self.ct_limbs, "c")
        m = _rows(m, self.n_limbs, "m")
        if c.shape[0] != m.shape[0]:
            raise ValueError("c and m batch sizes differ")
        out = np.empty_like(c)
        _check(lib().phe_hip_add_plain(self._h, _ptr(c), _ptr(m), _ptr(out), c.shape[0]))
        return out

    def powmod(self, base, exps):
        base = _rows(base, self.ct_limbs, "base")
        exps = np.ascontiguousarray(exps, dtype=np.uint32)
        if exps.ndim != 2 or exps.shape[0] != base.shape[0]:
            raise ValueError("exps must have shape (batch, exp_limbs)")
        out = np.empty_like(base)
        _check(lib().phe_hip_powmod(self._h, _ptr(base), _ptr(exps), exps.shape[1], _ptr(out), base.shape[0]))
        return out

    def multiexp(self, base, exps):
        """prod_i base[i]^exps[i] mod n^2 -> one row of ct_limbs words (1 for an empty batch)"""
        base = _rows(base, self.ct_limbs, "base")
        exps = np.ascontiguousarray(exps, dtype=np.uint32)
        if exps.ndim != 2 or exps.shape[0] != base.shape[0]:
            raise ValueError("exps must have shape (batch, exp_limbs)")
        out = np.empty((1, self.ct_limbs), dtype=np.uint32)
        _check(lib().phe_hip_multiexp(self._h, _ptr(base), _ptr(exps), max(1, exps.shape[1]), _ptr(out), base.shape[0]))
        return out

    # ---- decimal wire format (include/phe_hip.h "decimal wire format") ----
    @staticmethod
    def decimal_width(words):
        return lib().phe_hip_decimal_width(int(words))

    def to_decimal(self, limbs):
        """(rows, words) uint32 -> (rows, decimal_width(words)) uint8 ASCII digits, '0'-padded on the left"""
        limbs = np.ascontiguousarray(limbs, dtype=np.uint32)
        out = np.empty((limbs.shape[0], self.decimal_width(limbs.shape[1])), dtype=np.uint8)
        _check(lib().phe_hip_to_decimal(self._h, _ptr(limbs), limbs.shape[1], _ptr(out), out.shape[1], limbs.shape[0]))
        return out

    def to_decimal_dev(self, limbs_ptr, words, rows):
        """the same from a device-resident limb array: only the digits cross PCIe"""
        width = self.decimal_width(words)
        out = np.empty((rows, width), dtype=np.uint8)
        if rows == 0:
            return out
        nbytes = (rows * width + 3) // 4 * 4
        d = self.malloc(nbytes)
        try:
            _check(lib().phe_hip_to_decimal_dev(self._h, limbs_ptr, words, d, width, rows, 0))
            self.d2h(out, d)
        finally:
            self.free(d, nbytes)
        return out

    def from_decimal(self, digits, words):
        """(rows, width) uint8 ASCII digits (leading '0's allowed) -> (rows, words) uint32; ValueError like int()"""
        digits = np.ascontiguousarray(digits, dtype=np.uint8)
        out = np.empty((digits.shape[0], words), dtype=np.uint32)
        bad = ctypes.c_size_t(0)
        rc = lib().phe_hip_from_decimal(self._h, _ptr(digits), digits.shape[1], _ptr(out), words, digits.shape[0],
                                        ctypes.byref(bad))
        if rc == EINVAL:
            err = ValueError("row %d: %s" % (bad.value, lib().phe_hip_last_error().decode()))
            err.bad_index = bad.value
            raise err
        _check(rc)
        return out

    def invert(self, a):
        a = _rows(a, self.ct_limbs, "a")
        out = np.empty_like(a)
        bad = ctypes.c_size_t(0)
        rc = lib().phe_hip_invert(self._h, _ptr(a), _ptr(out), a.shape[0], ctypes.byref(bad))
        if rc != OK:
            _raise(rc, bad.value)
        return out

    # ---- device-pointer entry points (ints: device addresses; stream: hipStream_t as int) ----
    def encrypt_dev(self, m_ptr, r_ptr, c_ptr, batch, stream=0):
        _check(lib().phe_hip_encrypt_dev(self._h, m_ptr, r_ptr, c_ptr, batch, stream))

    def obfuscate_dev(self, cin_ptr, r_ptr, cout_ptr, batch, stream=0):
        _check(lib().phe_hip_obfuscate_dev(self._h, cin_ptr, r_ptr, cout_ptr, batch, stream))

    def decrypt_dev(self, c_ptr, m_ptr, batch, stream=0):
        _check(lib().phe_hip_decrypt_dev(self._h, c_ptr, m_ptr, batch, stream))

    def mulmod_dev(self, a_ptr, b_ptr, out_ptr, batch, stream=0):
        _check(lib().phe_hip_mulmod_dev(self._h, a_ptr, b_ptr, out_ptr, batch, stream))

    def montmul_dev(self, a_ptr, b_ptr, b_is_row, out_ptr, batch, stream=0):
        """out[i] = a[i] * b[i] / R mod n^2 (ONE Montgomery product; R = 2^mont_radix_bits()); b_is_row: one row b for all i"""
        _check(lib().phe_hip_montmul_dev(self._h, a_ptr, b_ptr, 1 if b_is_row else 0, out_ptr, batch, stream))

    def mont_radix_bits(self):
        bits = ctypes.c_int(0)
        _check(lib().phe_hip_mont_radix_bits(self._h, ctypes.byref(bits)))
        return bits.value

    def add_plain_dev(self, c_ptr, m_ptr, out_ptr, batch, stream=0):
        _check(lib().phe_hip_add_plain_dev(self._h, c_ptr, m_ptr, out_ptr, batch, stream))

    def powmod_dev(self, base_ptr, exp_ptr, exp_limbs, max_exp_bits, out_ptr, batch, stream=0):
        _check(lib().phe_hip_powmod_dev(self._h, base_ptr, exp_ptr, exp_limbs, max_exp_bits, out_ptr, batch, stream))

    def multiexp_dev(self, base_ptr, exp_ptr, exp_limbs, max_exp_bits, out_ptr, batch, stream=0):
        _check(lib().phe_hip_multiexp_dev(self._h, base_ptr, exp_ptr, exp_limbs, max_exp_bits, out_ptr, batch, stream))

    def multiexp_rows_dev(self, base_ptr, inv_ptr, exp_ptr, neg_ptr, exp_limbs, max_exp_bits, out_ptr, batch, rows, stream=0):
        _check(lib().phe_hip_multiexp_rows_dev(self._h, base_ptr, inv_ptr, exp_ptr, neg_ptr, exp_limbs, max_exp_bits,
                                               out_ptr, batch, rows, stream))

    def multiexp_csr_dev(self, base_ptr, inv_ptr, batch, row_ptr, cols_ptr, exp_ptr, neg_ptr, exp_limbs, max_exp_bits,
                         order_ptr, out_ptr, rows, stream=0):
        _check(lib().phe_hip_multiexp_csr_dev(self._h, base_ptr, inv_ptr, batch, row_ptr, cols_ptr, exp_ptr, neg_ptr,
                                              exp_limbs, max_exp_bits, order_ptr, out_ptr, rows, stream))

    def multiexp_csr(self, base, base_inv, row_ptr, cols, exps, neg, order, rows):
        """host arrays -> (rows, ct_limbs): the table-lookup multi-exponentiation (phe_hip_multiexp_csr_dev) staged through
        device blocks.  row_ptr / cols None = dense rows; exps: (entries, exp_limbs); neg: (entries,) bytes or None."""
        base = _rows(base, self.ct_limbs, "base")
        exps = np.ascontiguousarray(exps, dtype=np.uint32)
        out = np.empty((rows, self.ct_limbs), dtype=np.uint32)
        held = []

        def up(arr, dtype):
            if arr is None:
                return None
            arr = np.ascontiguousarray(arr, dtype=dtype)
            nbytes = max(4, (arr.nbytes + 3) // 4 * 4)
            p = self.malloc(nbytes)
            held.append((p, nbytes))
            if arr.nbytes:
                self.h2d(p, arr)
            return p
        try:
            b = up(base, np.uint32)
            bi = up(_rows(base_inv, self.ct_limbs, "base_inv"), np.uint32) if base_inv is not None else None
            rp, cl = up(row_ptr, np.uint64), up(cols, np.uint32)
            e, ng, od = up(exps, np.uint32), up(neg, np.uint8), up(order, np.uint32)
            o = self.malloc(max(4, out.nbytes))
            held.append((o, max(4, out.nbytes)))
            self.multiexp_csr_dev(b, bi, base.shape[0], rp, cl, e, ng, exps.shape[1], max_exp_bits(exps), od, o, rows)
            self.sync()
            if out.nbytes:
                self.d2h(out, o)
        finally:
            for p, nbytes in held:
                self.free(p, nbytes)
        return out

    def multiexp_rows(self, base, base_inv, exps, neg):
        """host arrays: base (batch, ct_limbs), base_inv same or None, exps (rows, batch, exp_limbs), neg (rows, batch)
        bytes or None -> (rows, ct_limbs).  Stages through device blocks (there is no host-pointer C entry for the
        matrix form)."""
        base = _rows(base, self.ct_limbs, "base")
        exps = np.ascontiguousarray(exps, dtype=np.uint32)
        rows, batch = exps.shape[0], base.shape[0]
        out = np.empty((rows, self.ct_limbs), dtype=np.uint32)
        held = []

        def up(arr):
            arr = np.ascontiguousarray(arr)
            nbytes = max(4, (arr.nbytes + 3) // 4 * 4)
            p = self.malloc(nbytes)
            held.append((p, nbytes))
            if arr.nbytes:
                self.h2d(p, arr)
            return p
        try:
            b = up(base)
            bi = up(_rows(base_inv, self.ct_limbs, "base_inv")) if base_inv is not None else None
            e = up(exps)
            ng = up(np.ascontiguousarray(neg, dtype=np.uint8)) if neg is not None else None
            o = self.malloc(max(4, out.nbytes))
            held.append((o, max(4, out.nbytes)))
            bits = max_exp_bits(exps)
            self.multiexp_rows_dev(b, bi, e, ng, exps.shape[2], bits, o, batch, rows)
            self.sync()
            if out.nbytes:
                self.d2h(out, o)
        finally:
            for p, nbytes in held:
                self.free(p, nbytes)
        return out

    def invert_dev(self, a_ptr, out_ptr, batch, stream=0):
        bad = ctypes.c_size_t(0)
        rc = lib().phe_hip_invert_dev(self._h, a_ptr, out_ptr, batch, ctypes.byref(bad), stream)
        if rc != OK:
            _raise(rc, bad.value)

    def select_rows_dev(self, a_ptr, b_ptr, mask_ptr, out_ptr, limbs, batch, stream=0):
        _check(lib().phe_hip_select_rows_dev(self._h, a_ptr, b_ptr, mask_ptr, out_ptr, limbs, batch, stream))

    def gather_rows_dev(self, src_ptr, src_rows, idx_ptr, dst_ptr, limbs, count, stream=0):
        """dst[j] = src[idx[j]] for `count` rows of `limbs` words (idx: uint32 row indices on the device; an index >= src_rows
        gives a row of zeros)"""
        _check(lib().phe_hip_gather_rows_dev(self._h, src_ptr, src_rows, idx_ptr, dst_ptr, limbs, count, stream))

    def scatter_rows_dev(self, src_ptr, idx_ptr, dst_ptr, dst_rows, limbs, count, stream=0):
        """dst[idx[j]] = src[j] (an index >= dst_rows is skipped)"""
        _check(lib().phe_hip_scatter_rows_dev(self._h, src_ptr, idx_ptr, dst_ptr, dst_rows, limbs, count, stream))

    def stream_create(self):
        """a stream whose launches overlap the blocking h2d / d2h copies (they are ordered on the NULL stream)"""
        p = ctypes.c_void_p(None)
        _check(lib().phe_hip_stream_create(self._h, ctypes.byref(p)))
        return p.value

    def stream_destroy(self, stream):
        if self._h and self._h.value and stream:
            lib().phe_hip_stream_destroy(self._h, stream)

    def sync(self, stream=0):
        _check(lib().phe_hip_stream_sync(self._h, stream))

    # ---- raw device memory (for hosts without a tensor library) ----
    # Freed blocks are kept for reuse by size (a hipMalloc/hipFree pair of a ciphertext vector costs more than the
    # homomorphic add that fills it).  Safe because every Python-level operation synchronises before it returns, so a
    # block is idle by the time its owner is garbage-collected.
    POOL_LIMIT = 4 << 30

    def malloc(self, nbytes):
        pool = self.__dict__.setdefault("_pool", {})
        blocks = pool.get(nbytes)
        if blocks:
            self._pooled -= nbytes
            return blocks.pop()
        p = ctypes.c_void_p(None)
        _check(lib().phe_hip_malloc(self._h, nbytes, ctypes.byref(p)))
        return p.value

    def free(self, ptr, nbytes=None):
        if not (self._h and self._h.value):
            return
        pool = self.__dict__.setdefault("_pool", {})
        pooled = self.__dict__.setdefault("_pooled", 0)
        if nbytes is not None and pooled + nbytes <= self.POOL_LIMIT:
            pool.setdefault(nbytes, []).append(ptr)
            self._pooled = pooled + nbytes
            return
        _check(lib().phe_hip_free(self._h, ptr))

    def trim_pool(self):
        for blocks in self.__dict__.get("_pool", {}).values():
            for ptr in blocks:
                lib().phe_hip_free(self._h, ptr)
        self._pool, self._pooled = {}, 0

    def h2d(self, dst_ptr, arr):
        arr = np.ascontiguousarray(arr)
        _check(lib().phe_hip_memcpy_h2d(self._h, dst_ptr, _ptr(arr), arr.nbytes))

    def d2h(self, arr, src_ptr):
        _check(lib().phe_hip_memcpy_d2h(self._h, _ptr(arr), src_ptr, arr.nbytes))

    def d2d(self, dst_ptr, src_ptr, nbytes, stream=0):
        _check(lib().phe_hip_memcpy_d2d(self._h, dst_ptr, src_ptr, nbytes, stream))


def comm_unique_id():
    """128 opaque bytes made by rank 0; every rank of the job passes the same bytes to Communicator"""
    buf = (ctypes.c_uint8 * 128)()
    _check(lib().phe_hip_comm_unique_id(buf))
    return bytes(buf)


class Communicator:
    """RCCL communicator inside the library (include/phe_hip.h "multi-GPU"): one per rank, on the context's device"""

    def __init__(self, ctx, unique_id, rank, world):
        self._h = ctypes.c_void_p(None)
        self.rank, self.world = rank, world
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(unique_id)
        _check(lib().phe_hip_comm_create(ctx._h, buf, rank, world, ctypes.byref(self._h)))

    def allgather_dev(self, local_ptr, all_ptr, rows, limbs, stream=0):
        _check(lib().phe_hip_allgather_dev(self._h, local_ptr, all_ptr, rows, limbs, stream))

    def close(self):
        if self._h and self._h.value:
            lib().phe_hip_comm_destroy(self._h)
            self._h = ctypes.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def miller_rabin(n, base, device=0):
    """pass[i] = n[i] is a strong probable prime to base[i]; (batch, limbs) uint32 rows -> bool array.
    One launch, one modulus per row (include/phe_hip.h "batched primality")."""
    n = np.ascontiguousarray(n, dtype=np.uint32)
    base = np.ascontiguousarray(base, dtype=np.uint32)
    if n.ndim != 2 or base.shape != n.shape:
        raise ValueError("n and base must be (batch, limbs) arrays of the same shape")
    out = np.zeros(n.shape[0], dtype=np.uint8)
    _check(lib().phe_hip_miller_rabin(int(device), _ptr(n), _ptr(base), n.shape[1], _ptr(out), n.shape[0]))
    return out.astype(bool)


def selftest_prims(device=0):
    out = np.zeros(1026, np.uint32)
    _check(lib().phe_hip_selftest_prims(device, _ptr(out)))
    return out
