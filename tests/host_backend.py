"""TEST INFRASTRUCTURE: a libgmp-backed stand-in for phe._native.Context, for tests of the HOST side that make thousands of
scalar calls (the reference's own example file, tests/test_reference_example_verbatim.py).

tests/emu_backend.EmuContext runs the device headers on a 64-fiber wave emulator: faithful, and ~0.1-0.8 s per one-row call.
This context answers the same calls from the checker (oracle/paillier_oracle: libgmp / CPython integers), so that what is under
test is the drop-in's Python layer — the object API, the encoding, the obfuscation state machine, the engine's argument
plumbing — at the speed of the reference itself.  Installed only by tests (monkeypatching phe._native.Context); the product has
no such path and fails without the HIP library (tests/test_abi_exports.py::test_no_cpu_fallback_without_library)."""
import numpy as np

from emu_backend import EmuContext, _emu_miller_rabin

_orc = None


def _oracle():
    global _orc
    if _orc is None:
        import os
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        if root not in sys.path:
            sys.path.insert(0, root)
        from oracle.paillier_oracle import COracle
        _orc = COracle()
    return _orc


class GmpContext(EmuContext):
    """the five hot calls from libgmp; everything else (multi-exponentiation, wire format, ...) stays with the emulator"""

    def info(self):
        return {"n_limbs": self.n_limbs, "ct_limbs": self.ct_limbs, "emulated": True, "stand_in": "libgmp"}

    def _ints(self, arr):
        return self._native.limbs_to_ints(np.ascontiguousarray(arr))

    def encrypt(self, m, r):
        if m.shape[0] == 0:
            return np.zeros((0, self.ct_limbs), np.uint32)
        return _oracle().encrypt(self._n_arr, np.ascontiguousarray(m), np.ascontiguousarray(r))

    def owner_encrypt_offered(self):
        return False

    def obfuscate(self, c_in, r):
        if c_in.shape[0] == 0:
            return c_in.copy()
        return _oracle().obfuscate(self._n_arr, np.ascontiguousarray(c_in), np.ascontiguousarray(r))

    def decrypt(self, c):
        if not self.has_private:
            raise ValueError("decrypt needs a private-key context")
        if c.shape[0] == 0:
            return np.zeros((0, self.n_limbs), np.uint32)
        p, q = self._key[0], self._key[1]
        return _oracle().decrypt(self._n_arr, p, q, np.ascontiguousarray(c))

    def mulmod(self, a, b):
        if a.shape[0] == 0:
            return a.copy()
        nsq = self.n * self.n
        return self._native.ints_to_limbs([x * y % nsq for x, y in zip(self._ints(a), self._ints(b))], self.ct_limbs)

    def add_plain(self, c, m):
        if c.shape[0] == 0:
            return c.copy()
        nsq = self.n * self.n
        return self._native.ints_to_limbs([x * (1 + self.n * y) % nsq for x, y in zip(self._ints(c), self._ints(m))], self.ct_limbs)

    def powmod(self, base, exps):
        if base.shape[0] == 0:
            return base.copy()
        nsq = self.n * self.n
        return self._native.ints_to_limbs([pow(x, e, nsq) for x, e in zip(self._ints(base), self._ints(exps))], self.ct_limbs)


def install(monkeypatch):
    from phe import _native
    monkeypatch.setattr(_native, "Context", GmpContext)
    monkeypatch.setattr(_native, "miller_rabin", _emu_miller_rabin)
