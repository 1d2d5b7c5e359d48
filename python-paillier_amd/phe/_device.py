"""Device-resident limb arrays: (rows, cols) uint32 blocks in HBM, owned through the C-ABI's memory helpers.

EncryptedVector(device=True) keeps its ciphertexts in one of these so that chains of homomorphic operations
(`+`, `*`, sum, dot, obfuscate, decrypt) never cross PCIe; only plaintext-sized data (scalars, exponents,
masks, decrypted mantissas) moves between host and device.
"""
import numpy as np


class DeviceArray:
    def __init__(self, ctx, rows, cols, _ptr=None, _owner=None):
        self.ctx = ctx
        self.rows, self.cols = int(rows), int(cols)
        self._owner = _owner                      # keeps the allocation alive for row views
        if _ptr is None:
            self._alloc = max(4, self.nbytes)
            self.ptr = ctx.malloc(self._alloc)
            self._owns = True
        else:
            self.ptr = _ptr
            self._owns = False

    @property
    def shape(self):
        return (self.rows, self.cols)

    @property
    def nbytes(self):
        return self.rows * self.cols * 4

    def __len__(self):
        return self.rows

    @classmethod
    def from_host(cls, ctx, arr, dtype=np.uint32):
        arr = np.ascontiguousarray(arr, dtype=dtype)
        if arr.ndim == 1:
            arr = arr.reshape(-1, 1)
        if dtype == np.uint8:                     # byte masks: stored packed, cols counts 4-byte words
            words = (arr.size + 3) // 4
            out = cls(ctx, words, 1)
            buf = np.zeros(words * 4, np.uint8)
            buf[:arr.size] = arr.ravel()
            ctx.h2d(out.ptr, buf)
            return out
        out = cls(ctx, arr.shape[0], arr.shape[1])
        if arr.size:
            ctx.h2d(out.ptr, arr)
        return out

    def to_host(self):
        out = np.empty((self.rows, self.cols), np.uint32)
        if out.size:
            self.ctx.d2h(out, self.ptr)
        return out

    def rows_view(self, lo, hi):
        """Rows [lo, hi) as a view (no copy); the parent allocation stays alive through the view."""
        lo, hi = max(0, lo), min(self.rows, hi)
        return DeviceArray(self.ctx, max(0, hi - lo), self.cols, _ptr=self.ptr + lo * self.cols * 4,
                           _owner=self._owner or self)

    def copy(self):
        out = DeviceArray(self.ctx, self.rows, self.cols)
        if self.nbytes:
            self.ctx.d2d(out.ptr, self.ptr, self.nbytes)
            self.ctx.sync()
        return out

    def __del__(self):
        try:
            if getattr(self, "_owns", False) and self.ptr:
                self.ctx.free(self.ptr, self._alloc)
                self.ptr = None
        except Exception:
            pass
