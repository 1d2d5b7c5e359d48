"""Single-process fan-out over several GPUs behind the drop-in API.

The reference's idiom is a plain loop in ONE process (examples/federated_learning_with_encryption.py:122-133,
examples/benchmarks.py:12-29); a user who switches to this package on an 8-GPU node and calls `pub.encrypt_batch(x)` /
`priv.decrypt_batch(v)` from one Python process should get all of them without launching torch.distributed.run (that form
is phe/sharding.py + bench.py: one process per GPU).  Opt-in:

    PHE_HIP_DEVICES=all          every visible device
    PHE_HIP_DEVICES=0,1,2,3      these devices (an id may repeat: two contexts on one GPU, each with its own streams)

A key then holds a FLEET: one Engine (one native context, one obfuscator pool, one lock) per listed device, the key's
ordinary engine first.  A batch is cut contiguously (`sharding.shard_bounds`, the same cut the multi-process path makes: every
element is independent, the key constants are tiny and replicated — SURVEY.md 8(e)) and every shard runs on its device from
a worker thread; ctypes drops the GIL inside each native call, so the devices work concurrently.  Results land in ONE host
array in shard order (`encrypt_batch`, `decrypt_batch`, host `+` / `*`), or stay resident as a per-device list
(`encrypt_batch_sharded`).  No data-path collective: the only exchange step of the path (the ciphertext concatenation) is a
host-side concatenate here.

Obfuscators made ahead of time (`precompute_obfuscators`) are per ENGINE, hence per device: a pool is never shared across
devices (an obfuscator is used once, by the device that holds it).
"""
import os
import threading
from concurrent.futures import ThreadPoolExecutor

from .sharding import shard_bounds

# a shard smaller than this is not worth a second device: the launch is latency-bound anyway
MIN_ROWS_PER_DEVICE = int(os.environ.get("PHE_HIP_FLEET_MIN_ROWS", "2048"))


def configured_devices():
    """The device list of PHE_HIP_DEVICES, or None when no fan-out is asked for (unset, one device, one visible GPU)."""
    v = os.environ.get("PHE_HIP_DEVICES", "").strip()
    if not v:
        return None
    if v.lower() == "all":
        from . import _native
        try:
            count = _native.device_count()
        except Exception:
            return None
        return list(range(count)) if count > 1 else None
    try:
        ids = [int(x) for x in v.split(",") if x.strip() != ""]
    except ValueError:
        raise ValueError("PHE_HIP_DEVICES must be 'all' or a comma-separated list of device ids, not %r" % v)
    return ids if len(ids) > 1 else None


class Fleet:
    """One engine per listed device; engines[0] is the key's ordinary engine.  The others are made on first use (a context
    per device costs a few MB of constants and ~0.1 s) by `make_engine(device)`."""

    def __init__(self, primary, make_engine, devices):
        self.devices = list(devices)
        self._engines = [primary] + [None] * (len(self.devices) - 1)
        self._make = make_engine
        self._lock = threading.Lock()
        self._pool = ThreadPoolExecutor(max_workers=len(self.devices), thread_name_prefix="phe-fleet")
        self._retired = []     # (native context of an engine of a fleet this one replaced, its slot here): resident rows made there resolve to the successor
        self._inherit = {}     # slot -> obfuscator pool object of the replaced fleet's engine there: adopted when the successor is made

    def __len__(self):
        return len(self.devices)

    def engine(self, k):
        eng = self._engines[k]
        if eng is None:
            with self._lock:
                if self._engines[k] is None:
                    made = self._make(self.devices[k])
                    if k in self._inherit:                     # the pool OBJECT of the engine this one succeeds (see succeed)
                        made._obf = self._inherit.pop(k)
                    self._engines[k] = made
                eng = self._engines[k]
        return eng

    def engines(self):
        return [self.engine(k) for k in range(len(self.devices))]

    def engine_of(self, ctx):
        """the fleet's engine that owns the native context `ctx` (a resident vector's home) — or, for a context of a fleet this
        one replaced (succeed), the engine that took its place on the SAME device —, or None"""
        for eng in self._engines:
            if eng is not None and eng.ctx is ctx:
                return eng
        for old_ctx, k in self._retired:
            if old_ctx is ctx:
                return self.engine(k)
        return None

    def engine_on(self, device):
        """an engine of the fleet on `device` (device pointers are valid across the contexts of one GPU), or None"""
        for k, d in enumerate(self.devices):
            if d == device:
                return self.engine(k)
        return None

    def succeed(self, old):
        """Take the place of the fleet `old` (same device list): every engine `old` had made gets a successor here on the same
        device, with the same obfuscator pool OBJECT (rows are taken under one lock whichever engine a thread is in), and
        resident rows made by `old`'s engines resolve to those successors (engine_of).  The key pair's engines replacing a
        public key's (PaillierPrivateKey._get_engine): a vector that encrypt_batch_sharded left on device k must be decrypted
        by an engine ON device k that holds the private key — never by the device-0 engine.

        The successors are made on FIRST USE (engine(k) adopts the pool then): the caller holds the key's engine lock, and a
        context per device at ~0.1 s each would stall every thread waiting on it.  Of a retired engine only its native context
        is kept — resident rows name it as their home and free their memory through it —, with its window tables and scratch
        rows given back to the device; the replaced fleet's worker threads are told to exit."""
        assert list(old.devices) == self.devices
        for k, eng in enumerate(old._engines):
            if eng is None:
                continue
            if k > 0:
                if self._engines[k] is not None:
                    self._engines[k]._obf = eng._obf
                else:
                    self._inherit[k] = eng._obf
            if eng is not self._engines[k]:
                self._retired.append((eng.ctx, k))
                try:
                    eng.ctx.release_scratch()
                except Exception:  # noqa: BLE001 — a backend without scratch to give back (the emulator)
                    pass
        self._retired.extend(old._retired)
        self._inherit.update({k: v for k, v in old._inherit.items() if self._engines[k] is None and k not in self._inherit})
        if old._pool is not self._pool:
            old._pool.shutdown(wait=False)
        return self

    def shards(self, rows, min_rows=None):
        """contiguous [lo, hi) bounds, one per device that gets work (at least `min_rows` rows each; one shard = no fan-out)"""
        min_rows = MIN_ROWS_PER_DEVICE if min_rows is None else min_rows
        k = max(1, min(len(self.devices), rows // max(1, min_rows)))
        return [shard_bounds(rows, k, r) for r in range(k)]

    def run(self, rows, fn, min_rows=None):
        """[fn(engine_k, lo_k, hi_k) for every shard k], the shards on worker threads, results in shard order.  The first
        exception of any shard is raised after all of them have finished (no shard is left running on a device)."""
        bounds = self.shards(rows, min_rows)
        if len(bounds) == 1:
            return [fn(self.engine(0), 0, rows)]
        futures = [self._pool.submit(fn, self.engine(k), lo, hi) for k, (lo, hi) in enumerate(bounds)]
        results, first_error = [], None
        for f in futures:
            try:
                results.append(f.result())
            except BaseException as e:  # noqa: BLE001 — re-raised below, once every shard is done
                first_error = first_error or e
                results.append(None)
        if first_error is not None:
            raise first_error
        return results

    def each(self, fn):
        """[fn(engine_k, k) for every device], concurrently (e.g. filling every device's obfuscator pool with its share)"""
        futures = [self._pool.submit(fn, self.engine(k), k) for k in range(len(self.devices))]
        return [f.result() for f in futures]

    def made(self):
        """the engines that exist so far"""
        return [eng for eng in self._engines if eng is not None]
