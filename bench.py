#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on the config it is quoted on, plus the other configs' legs.

Metric : "Paillier 2048-bit encrypts/sec + decrypts/sec per node (bit-exact)"
Config : configs[1] — 2048-bit key, 1M-plaintext batch encrypt + decrypt on 1 MI355X (per GPU; weak scaling).

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU.  Started by `torch.distributed.run` (RANK / LOCAL_RANK / WORLD_SIZE in the environment) the
script is a rank; started bare with --gpus N > 1 it re-launches ITSELF under `torch.distributed.run --nproc-per-node N`
(127.0.0.1 rendezvous, backend nccl = RCCL), so both ways of calling it give N ranks and `n_gpus: N`.

A "step" is one pass of raw_encrypt over the batch (`value` = encrypts/s, whole job); the same K steps of raw_decrypt
are timed as a second region ("decrypt").  Operands are resident in HBM before every timed region.  Ranks only meet
at the barriers (the batch is embarrassingly sharded: no data-path collective in configs[1]).

The same JSON line also carries
  ops          — configs[2]: _raw_add, _raw_mul by float-like 56-bit / int64 scalars and with 10 % negative scalars
                 (the inverse branch of phe/paillier.py:745-749), obfuscate — each over the whole batch, with its own
                 roofline entry and a strided sample checked against the libgmp oracle;
  config4      — configs[3]: a 3072-bit key, one shard of 2^20 plaintexts per GPU (8M on 8 GPUs; at N = 1 that one shard;
                 --scaling strong: the 8M job itself cut over the ranks present), the ciphertext shards concatenated on every GPU
                 by ONE RCCL all-gather (phe.sharding.all_gather_rows)
                 and, for N > 1, once more by the library's own RCCL communicator (phe_hip_allgather_dev), bits compared — as the
                 LAST leg, under a watchdog (--lib-allgather-timeout): if it hangs the line goes out without it;
                 shard-boundary rows + a strided sample against the oracle;
  roofline     — dominant kernel (k_modexp_split<4,18,encrypt>): `frac` = multiply-adds the kernel EXECUTES
                 (v_mad_u64_u32 lane-operations, exact count from profiles/executed_mads_r*.json, cross-checked with the
                 PMC SQ_INSTS_VALU of profiles/) per second / integer-VALU peak — a fraction of the hardware limit, <= 1.
                 `canonical_frac` prices the same time with SURVEY.md 8(d)'s algorithmic MAC32 count (full-width
                 Montgomery): the split-modulus kernels need fewer multiplies than that, so it exceeds `frac`.
  cpu_baseline — the libgmp oracle (what gmpy2 executes) on all host cores over a bounded sample, rank 0 at N=1 only.

`--selftest-emu` (CPU contract test only, tests/test_bench_contract.py): gloo + the CPU wave emulator of tests/emu on a
256-bit key and a handful of rows, to check the launch / sharding / JSON plumbing without a GPU.  Not a measurement.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "python-paillier_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

METRIC = "Paillier 2048-bit encrypts/sec + decrypts/sec per node (bit-exact)"


# ---- bookkeeping (checked on CPU by tests/test_bench_contract.py) -------------------------------------------------
def _E(t):
    """modmuls of a t-bit exponentiation, canonical (t squarings, w = 5 sliding window, 16-entry table)"""
    return t + -(-t // 6) + 16


def _mont(s):
    return 2 * s * s + s


def mac32_counts(key_bits):
    """Algorithmic multiply-accumulates per encrypt / decrypt, exactly SURVEY.md 8(d)."""
    s2, s1, sh = key_bits // 16, key_bits // 32, key_bits // 64
    enc = s1 * s1 + (_E(key_bits) + 3) * _mont(s2)
    dec = 2 * _mont(s1) + 2 * (_E(key_bits // 2) + 2) * _mont(s1) + 3 * sh * sh + 4 * _mont(sh)
    return enc, dec


def mac32_ops(key_bits):
    """SURVEY.md 8(d) for configs[2]: add(k) = 2 s2^2, mul_t(k) = (E(t)+2) MontMul(s2); obfuscate = encrypt's
    exponentiation with one more product in place of the s1^2 plaintext term."""
    s2 = key_bits // 16
    return {"raw_add": 2 * s2 * s2, "raw_mul_float56": (_E(56) + 2) * _mont(s2), "raw_mul_int64": (_E(63) + 2) * _mont(s2),
            "obfuscate": (_E(key_bits) + 4) * _mont(s2)}


def executed_mads(key_bits, info):
    """MODEL of the v_mad_u64_u32 lane-operations one encrypt / one decrypt executes (29-bit limbs; t squarings,
    ceil(t/(w+1)) + 2^(w-1) products for the window w in use).  Split engine: a squaring is (3 + s/L) H^2 with s = L/2 + 1
    (even L) or (L + 1)/2 (odd L) limbs per lane and row in the symmetric first word (csrc/split_core.h sq_row; 4 H^2 before
    round 5), a product 5 H^2, entry 4 H^2 per input chunk (+5 H^2 when more than one), exit ~10 H^2; full-width engine:
    2 S^2 per Montgomery product.  Only the fallback when profiles/executed_mads_r*.json has no exact count for the geometry."""
    E_sq = lambda t: t
    # sliding windows of w = 6 bits (32 odd powers) from ~1000-bit exponents on, w = 5 (16) below: key_setup.h:pick_window
    E_mul = lambda t: (-(-t // 7) + 32) if t > 900 else (-(-t // 6) + 16)

    def modexp(t, lane_limbs, split, in_bits, extra):
        G, L = divmod(lane_limbs, 100)
        H = G * L
        if split:
            chunks = max(1, -(-in_bits // (29 * H)))
            sq = 3 + (L // 2 + 1 if L % 2 == 0 else (L + 1) // 2) / L
            return int(round((sq * (E_sq(t) + 1) - 4 + 5 * E_mul(t) + 4 * chunks + (5 if chunks > 1 else 0) + 10 + extra) * H * H))
        return (E_sq(t) + E_mul(t) + 3) * 2 * H * H

    enc = modexp(key_bits, info["lane_limbs_pub"], info["engine_pub"] == "split", key_bits, 2)
    dec = 2 * modexp(key_bits // 2, info["lane_limbs_priv"], info["engine_priv"] == "split", 2 * key_bits, 0)
    return enc, dec


def current_csrc_hash():
    """hash of the device sources in the tree (tools/csrc_hash.py): committed counts made from other sources are STALE"""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from csrc_hash import csrc_hash
        return csrc_hash()
    except Exception:
        return None
    finally:
        if sys.path and sys.path[0] == os.path.join(ROOT, "tools"):
            sys.path.pop(0)


def committed_count_is_stale(basename):
    """True when profiles/<basename> was made from other device sources than the tree holds (or does not say which)"""
    if not basename:
        return None
    try:
        with open(os.path.join(ROOT, "profiles", basename)) as f:
            made_from = json.load(f).get("csrc_sha256")
    except Exception:
        return True
    return made_from is None or made_from != current_csrc_hash()


def counted_mads(key_bits, info=None):
    """EXACT executed multiply-adds per element from the newest profiles/executed_mads_r*.json (tools/count_executed_mads.py:
    every wave::mad64 call of the device headers counted by the CPU wave emulator on full wavefronts) -> (dict, file),
    or (None, None) when there is no count for this key size / geometry."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "executed_mads_r*.json")))
    for path in reversed(files):
        try:
            with open(path) as f:
                rec = json.load(f)["keys"][str(key_bits)]
        except Exception:
            continue
        if info is not None and info.get("engine_pub") == "split":
            if list(divmod(info["lane_limbs_pub"], 100)) != list(rec.get("split_geometry_GL", [])):
                continue
        elif info is not None and not info.get("emulated"):
            continue
        return rec, os.path.basename(path)
    return None, None


def valu_peak_mac32(n_cus=256, clock_hz=2.4e9):
    """Integer-VALU peak in MAC32/s.

    `peak` is the nominal issue limit: v_mad_u64_u32 is a half-rate VALU op on gfx950 (4 cycles per wave64
    instruction per SIMD — calibrated by csrc/microbench.hip, NOT the quarter-rate of a 4-cycle base that
    SURVEY.md 8(d) assumed), so peak = CUs x 4 SIMD x 64 lanes x clock / 4 = 39.3e12 at 2.4 GHz.
    `sustained` is what a pure v_mad_u64_u32 stream actually reached in the newest committed microbenchmark run
    (8 waves/SIMD, 8 independent chains), for reference."""
    peak = n_cus * 4 * 64 * clock_hz / 4.0
    sustained, src = None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "microbench_r*.json")))
    if files:
        try:
            with open(files[-1]) as f:
                mb = json.load(f)
            sustained = float(mb["tests"]["v_mad_u64_u32"]["lane_ops_per_s"])
            src = os.path.basename(files[-1])
        except Exception:
            pass
    return peak, sustained, src


def measured_traffic_per_unit(kernel_key):
    """HBM bytes per ciphertext from the newest committed PMC run (profiles/hbm_traffic_r*.json), or None."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "hbm_traffic_r*.json")))
    for path in reversed(files):
        try:
            with open(path) as f:
                t = json.load(f)[kernel_key]
            return (t["fetch_bytes"] + t["write_bytes"]) / float(t["batch"]), os.path.basename(path)
        except Exception:
            continue
    return None, None


def pmc_valu_per_unit(kernel_key):
    """SQ_INSTS_VALU wave-instructions per element of the newest committed PMC summary that has them -> (value, file)"""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "hbm_traffic_r*.json")))
    for path in reversed(files):
        try:
            with open(path) as f:
                t = json.load(f)[kernel_key]
            return float(t["sq_insts_valu"]) / float(t["batch"]), os.path.basename(path)
        except Exception:
            continue
    return None, None


def measured_ops_traffic(key_bits):
    """PMC records of the kernels behind the `ops` entries (configs[2]) from the newest committed hbm_traffic_r*.json that has them:
    {op name: {"bytes_per_row", "valu_wave_instructions_per_row", "kernel", ...}}, file — tools/gpu_pmc_traffic.sh, legs ops<bits>"""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "hbm_traffic_r*.json")))
    for path in reversed(files):
        try:
            with open(path) as f:
                recs = json.load(f)["ops"][str(key_bits)]
            return recs, os.path.basename(path)
        except Exception:
            continue
    return {}, None


def host_cores():
    """Usable host cores: CPU affinity, further limited by a cgroup CPU quota if one is set."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return n


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1 << 20, help="plaintexts per GPU per step")
    ap.add_argument("--key-bits", type=int, default=2048, choices=[256, 1024, 2048, 3072])
    ap.add_argument("--cpu-sample", type=int, default=0, help="elements for the CPU baseline (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ops", action="store_true", help="skip the configs[2] leg")
    ap.add_argument("--only", choices=["encrypt", "decrypt"], default=None,
                    help="profiling aid (tools/gpu_profile_round.sh): time just this region and print a short line, so that a "
                         "rocprofv3 kernel trace of the run holds one leg's launches only; not a bench line")
    ap.add_argument("--config4", action="store_true", help="(kept for old command lines: the configs[3] leg now always runs)")
    ap.add_argument("--no-config4", action="store_true")
    ap.add_argument("--config4-key-bits", type=int, default=3072, choices=[256, 1024, 2048, 3072])
    ap.add_argument("--config4-total", type=int, default=0,
                    help="plaintexts of the whole configs[3] job (0 = weak: 2^20 per GPU, the per-GPU shard of the 8M job on 8 "
                         "GPUs; strong: 2^23 at every N)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): --batch plaintexts PER GPU and a 2^20-row configs[3] shard per GPU, whatever N; "
                         "strong: --batch is the WHOLE job, sharded contiguously over the ranks, and configs[3] is the 8M job "
                         "(2^23 rows) at every N.  The JSON line says which one it is (`scaling`).")
    ap.add_argument("--lib-allgather", action="store_true",
                    help="N = 1: repeat the configs[3] gather through the library's own RCCL communicator (always on for N > 1)")
    ap.add_argument("--no-lib-allgather", action="store_true")
    ap.add_argument("--lib-allgather-timeout", type=float, default=240.0,
                    help="seconds the library's own RCCL gather may take (communicator set-up included) before the line is printed "
                         "without it: a second RCCL instance that hangs must not cost the run its line")
    ap.add_argument("--oracle-sample", type=int, default=4096, help="strided rows of the timed batch checked against libgmp")
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--selftest-emu", action="store_true", help="CPU contract test: gloo + wave emulator, not a measurement")
    ap.add_argument("--ops-sample", type=int, default=4096, help="strided rows of every configs[2] result checked against libgmp")
    ap.add_argument("--config4-sample", type=int, default=4096,
                    help="strided rows of the configs[3] job checked against libgmp (the shard boundaries are always checked)")
    ap.add_argument("--inject-fault", choices=["cpu_baseline", "raw_add", "config4", "lib_allgather"], default=None,
                    help="TEST HOOK (tests/test_bench_contract.py): flip one bit of one result row of that leg after it was computed "
                         "and before it is checked — the run must then exit non-zero")
    return ap.parse_args(argv)


def relaunch_under_torchrun(args):
    """`bench.py --gpus N` started bare: become N ranks (one per GPU) by re-launching under torch.distributed.run."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on these hosts (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


# ---- the two backends: where operands live and how a launch is issued / timed -----------------------------------------
class HipBackend:
    """The product path: operands are torch tensors in HBM (torch = device memory + streams + torch.distributed, nothing
    else), every operation is one call through the C-ABI's device-pointer entry points on torch's current stream."""
    name, dist_backend = "hip", "nccl"

    def __init__(self, local_rank):
        import torch
        self.torch = torch
        torch.cuda.set_device(local_rank)
        self.dev = torch.device("cuda", local_rank)
        self.local_rank = local_rank
        self.stream = torch.cuda.current_stream().cuda_stream

    def context(self, n, p=None, q=None, hp=None, hq=None, pinv=None, n_limbs=None):
        from phe import _native as native
        return native.Context(n, p, q, hp, hq, pinv, device=self.local_rank, n_limbs=n_limbs)

    def rand(self, rows, cols, seed):
        gen = self.torch.Generator(device=self.dev)
        gen.manual_seed(seed)
        return self.torch.randint(-2 ** 31, 2 ** 31, (rows, cols), dtype=self.torch.int32, device=self.dev, generator=gen)

    def empty(self, rows, cols):
        return self.torch.empty((rows, cols), dtype=self.torch.int32, device=self.dev)

    def cat(self, parts):
        return self.torch.cat(parts).contiguous()

    def roll(self, t):
        return self.torch.roll(t, 1, 0).contiguous()

    def mask_every(self, rows, step):
        m = self.torch.zeros(rows, dtype=self.torch.uint8, device=self.dev)
        m[::step] = 1
        return m

    def np(self, t):
        import numpy as np
        return t.cpu().numpy().view(np.uint32) if t.dtype == self.torch.int32 else t.cpu().numpy()

    def take(self, t, idx):
        return t[self.torch.as_tensor(idx, device=self.dev, dtype=self.torch.long)]

    def equal(self, a, b):
        return bool(self.torch.equal(a, b))

    def sync(self):
        self.torch.cuda.synchronize()

    def as_tensor(self, t):
        return t

    def scalar_tensor(self, values):
        return self.torch.tensor(values, dtype=self.torch.float64, device=self.dev)

    def events(self, k):
        E = self.torch.cuda.Event
        return [(E(enable_timing=True), E(enable_timing=True)) for _ in range(k)]

    def record(self, ev):
        ev.record()

    def elapsed_ms(self, a, b):
        return a.elapsed_time(b)

    # one launch each (device pointers + torch's current stream)
    def encrypt(self, ctx, m, r, c, rows):
        ctx.encrypt_dev(m.data_ptr(), r.data_ptr(), c.data_ptr(), rows, self.stream)

    def decrypt(self, ctx, c, m, rows):
        ctx.decrypt_dev(c.data_ptr(), m.data_ptr(), rows, self.stream)

    def mulmod(self, ctx, a, b, out, rows):
        ctx.mulmod_dev(a.data_ptr(), b.data_ptr(), out.data_ptr(), rows, self.stream)

    def powmod(self, ctx, base, e, bits, out, rows):
        ctx.powmod_dev(base.data_ptr(), e.data_ptr(), e.shape[1], bits, out.data_ptr(), rows, self.stream)

    def montmul(self, ctx, a, b, b_is_row, out, rows):
        ctx.montmul_dev(a.data_ptr(), b.data_ptr(), b_is_row, out.data_ptr(), rows, self.stream)

    def encrypt_owner(self, ctx, m, r, c, rows):
        ctx.encrypt_owner_dev(m.data_ptr(), r.data_ptr(), c.data_ptr(), rows, self.stream)

    # resident rows in the pair form (include/phe_hip.h "pair form")
    def to_pair(self, ctx, c, pair, rows):
        ctx.to_pair_dev(c.data_ptr(), pair.data_ptr(), rows, self.stream)

    def pair_mul(self, ctx, a, b, out, rows):
        ctx.pair_mul_dev(a.data_ptr(), b.data_ptr(), False, out.data_ptr(), rows, self.stream)

    def from_pair(self, ctx, pair, out, rows):
        ctx.from_pair_dev(pair.data_ptr(), None, out.data_ptr(), rows, self.stream)

    def pair_powmod(self, ctx, a, e, bits, out, rows):
        ctx.pair_powmod_dev(a.data_ptr(), e.data_ptr(), e.shape[1], bits, out.data_ptr(), rows, self.stream)

    def upload(self, arr):
        import numpy as np
        return self.torch.from_numpy(np.ascontiguousarray(arr).view(np.int32)).to(self.dev)

    def obfuscate(self, ctx, c, r, out, rows):
        ctx.obfuscate_dev(c.data_ptr(), r.data_ptr(), out.data_ptr(), rows, self.stream)

    def invert(self, ctx, a, out, rows):
        ctx.invert_dev(a.data_ptr(), out.data_ptr(), rows, self.stream)

    def select_rows(self, ctx, a, b, mask, out, rows):
        ctx.select_rows_dev(a.data_ptr(), b.data_ptr(), mask.data_ptr(), out.data_ptr(), a.shape[1], rows, self.stream)

    def index_every(self, rows, step):
        """row indices 0, step, 2 step ... < rows as a device array of 32-bit words"""
        return self.torch.arange(0, rows, step, dtype=self.torch.int32, device=self.dev)

    def gather_rows(self, ctx, src, idx, dst, count):
        ctx.gather_rows_dev(src.data_ptr(), src.shape[0], idx.data_ptr(), dst.data_ptr(), src.shape[1], count, self.stream)

    def scatter_rows(self, ctx, src, idx, dst, count):
        ctx.scatter_rows_dev(src.data_ptr(), idx.data_ptr(), dst.data_ptr(), dst.shape[0], src.shape[1], count, self.stream)

    def copy_rows(self, dst, src, rows):
        dst[:rows].copy_(src[:rows])


class EmuSelftestBackend:
    """CPU contract test only (--selftest-emu): numpy operands, the wave emulator of tests/emu behind the same method
    names, wall-clock 'events'.  Exists so that the launch / sharding / JSON plumbing is testable without a GPU."""
    name, dist_backend = "selftest-emu", "gloo"

    def __init__(self, local_rank):
        import numpy as np
        self.npmod = np
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import emu_backend
        self.EmuContext = emu_backend.EmuContext
        self.local_rank = local_rank

    def context(self, n, p=None, q=None, hp=None, hq=None, pinv=None, n_limbs=None):
        return self.EmuContext(n, p, q, hp, hq, pinv, n_limbs=n_limbs)

    def rand(self, rows, cols, seed):
        rng = self.npmod.random.Generator(self.npmod.random.PCG64(seed))
        return rng.integers(0, 2 ** 32, (rows, cols), dtype=self.npmod.uint64).astype(self.npmod.uint32)

    def empty(self, rows, cols):
        return self.npmod.zeros((rows, cols), dtype=self.npmod.uint32)

    def cat(self, parts):
        return self.npmod.ascontiguousarray(self.npmod.concatenate(parts))

    def roll(self, t):
        return self.npmod.ascontiguousarray(self.npmod.roll(t, 1, 0))

    def mask_every(self, rows, step):
        m = self.npmod.zeros(rows, dtype=self.npmod.uint8)
        m[::step] = 1
        return m

    def np(self, t):
        return t

    def take(self, t, idx):
        return self.npmod.ascontiguousarray(t[self.npmod.asarray(idx, dtype=self.npmod.int64)])

    def equal(self, a, b):
        return bool(self.npmod.array_equal(a, b))

    def sync(self):
        pass

    def as_tensor(self, t):
        import torch
        return torch.from_numpy(self.npmod.ascontiguousarray(t).view(self.npmod.int32))

    def scalar_tensor(self, values):
        import torch
        return torch.tensor(values, dtype=torch.float64)

    def events(self, k):
        return [([0.0], [0.0]) for _ in range(k)]

    def record(self, ev):
        ev[0] = time.perf_counter()

    def elapsed_ms(self, a, b):
        return (b[0] - a[0]) * 1e3

    def encrypt(self, ctx, m, r, c, rows):
        c[:rows] = ctx.encrypt(m[:rows], r[:rows])

    def decrypt(self, ctx, c, m, rows):
        m[:rows] = ctx.decrypt(c[:rows])

    def mulmod(self, ctx, a, b, out, rows):
        out[:rows] = ctx.mulmod(a[:rows], b[:rows])

    def powmod(self, ctx, base, e, bits, out, rows):
        out[:rows] = ctx.powmod(base[:rows], e[:rows])

    def obfuscate(self, ctx, c, r, out, rows):
        out[:rows] = ctx.obfuscate(c[:rows], r[:rows])

    def invert(self, ctx, a, out, rows):
        out[:rows] = ctx.invert(a[:rows])

    def select_rows(self, ctx, a, b, mask, out, rows):
        out[:rows] = self.npmod.where(mask[:rows, None] != 0, b[:rows], a[:rows])


def config4_memory_budget(rows, total, t1, t2, gathered, lib_gathered):
    """Bytes the configs[3] leg allocates on ONE GPU: `rows` of the `total`-row job live here (operands m, r of t1 words, the
    ciphertext shard of t2 words), the gathered vector once per all-gather form that runs, the engine's window tables and
    scratch (an upper bound: 576 B x 32 entries per resident limb group, a few hundred MB at most).
    tests/test_bench_contract.py holds the real 3072-bit / 8M job on 8 ranks against the 288 GB of an MI355X with it."""
    return {"operands_m_r": 2 * rows * t1 * 4, "ciphertext_shard": rows * t2 * 4,
            "gathered_vector_torch": (total * t2 * 4) if gathered else 0,
            "gathered_vector_library_rccl": (total * t2 * 4) if lib_gathered else 0,
            "window_tables_and_scratch_upper_bound": 2 << 30}


def clamp_operands(m, r, s1, n_int):
    """random words -> m < n, 1 <= r < n (n has exactly 32*s1 bits in every fixture key)"""
    top = (n_int >> (32 * (s1 - 1)))                     # the top word of n: >= 2^31
    assert top >= 1 << 31
    m[:, s1 - 1] = 0
    r[:, s1 - 1] &= 0x3fffffff
    r[:, 0] |= 1


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_under_torchrun(args)

    # stdout carries exactly ONE line (rank 0's JSON): libraries that print on their own (RCCL's version banner when a
    # communicator comes up) are sent to stderr for the whole run; the line itself is written to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch.distributed as dist
    from phe import _native as native
    from phe.sharding import all_gather_rows, shard_bounds

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: the launcher's world size is used" % (args.gpus, world), file=sys.stderr)
    use_dist = world > 1 or "RANK" in os.environ          # launched by torch.distributed.run (also with 1 rank)
    be = (EmuSelftestBackend if args.selftest_emu else HipBackend)(local_rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if be.name == "hip":
            dist.init_process_group(backend="nccl", device_id=be.dev)
        else:
            dist.init_process_group(backend="gloo")

    def golden(bits):
        with open(os.path.join(ROOT, "tests", "golden", "paillier_%d.json" % bits)) as f:
            g = json.load(f)
        return {k: int(g[k], 16) for k in ("n", "p", "q", "hp", "hq", "p_inverse")}

    key = golden(args.key_bits)
    n_int = key["n"]
    s1, s2 = args.key_bits // 32, args.key_bits // 16
    ctx = be.context(n_int, key["p"], key["q"], key["hp"], key["hq"], key["p_inverse"], n_limbs=s1)
    if args.blocks_per_cu and hasattr(ctx, "set_blocks_per_cu"):
        ctx.set_blocks_per_cu(args.blocks_per_cu)

    # ---- synthetic inputs, generated on the device (resident in HBM before any timed region) ----
    strong = args.scaling == "strong"
    if strong:                                                 # --batch is the whole job: this rank's contiguous shard of it
        lo_b, hi_b = shard_bounds(args.batch, world, rank)
        B = hi_b - lo_b
        if B == 0:
            raise SystemExit("bench.py --scaling strong: --batch %d leaves rank %d of %d without a row" % (args.batch, rank, world))
    else:
        B = args.batch
    job_rows = args.batch if strong else world * B             # rows of the whole job per step
    m = be.rand(B, s1, 1234 + 2 * rank)
    r = be.rand(B, s1, 1235 + 2 * rank)
    clamp_operands(m, r, s1, n_int)
    c = be.empty(B, s2)
    m_back = be.empty(B, s1)

    def barrier():
        be.sync()
        if use_dist:
            dist.barrier()

    def max_over_ranks(values):
        if not use_dist:
            return list(values)
        t = be.scalar_tensor(list(values))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(x) for x in t.tolist()]

    def timed(step_fn, steps):
        """K steps bracketed by barrier + synchronize on both sides (max over ranks); also per-launch event durations."""
        evs = be.events(steps)
        barrier()
        t0 = time.perf_counter()
        for a, b in evs:
            be.record(a)
            step_fn()
            be.record(b)
        barrier()
        dt = max_over_ranks([time.perf_counter() - t0])[0]
        return dt, [be.elapsed_ms(a, b) for a, b in evs]

    enc_step = lambda: be.encrypt(ctx, m, r, c, B)
    dec_step = lambda: be.decrypt(ctx, c, m_back, B)
    if args.only:
        enc_step()                                              # (the decrypt leg needs ciphertexts; one launch, outside the timed region)
        step = enc_step if args.only == "encrypt" else dec_step
        for _ in range(args.warmup):
            step()
        dt, launch_ms = timed(step, args.steps)
        if rank == 0:
            print(json.dumps({"profiling_aid_not_a_bench_line": True, "only": args.only, "batch": B, "key_bits": args.key_bits,
                              "steps": args.steps, "per_s": job_rows * args.steps / dt, "launch_ms": launch_ms,
                              "decrypt_round_trip": be.equal(m_back, m) if args.only == "decrypt" else None}))
        return
    for _ in range(args.warmup):
        enc_step()
    enc_dt, enc_launch_ms = timed(enc_step, args.steps)
    for _ in range(args.warmup):
        dec_step()
    dec_dt, dec_launch_ms = timed(dec_step, args.steps)
    try:  # the rung the timed decrypt really ran on (the ladder need not end on the context's narrowest geometry)
        dec_geom = int(ctx.last_launch()["geom_priv"]) if hasattr(ctx, "last_launch") else 0
    except Exception:
        dec_geom = 0

    # ---- bit-exactness of what was just timed -------------------------------------------------
    roundtrip_ok = be.equal(m_back, m)
    cores = host_cores()
    orc = None
    if rank == 0:
        from oracle.paillier_oracle import COracle             # the CHECKER (and the cpu_baseline leg), never the product
        orc = COracle()
    n_arr = native.int_to_limbs(n_int, s1)

    def strided(rows, count):
        return sorted(set(range(0, rows, max(1, rows // max(1, count)))))[:count]

    sample_ok, sample_rows = None, 0
    if rank == 0:
        idx = strided(B, args.oracle_sample)
        want = orc.encrypt(n_arr, be.np(be.take(m, idx)), be.np(be.take(r, idx)), nthreads=cores)
        sample_ok = bool(np.array_equal(be.np(be.take(c, idx)), want))
        sample_rows = len(idx)

    # ---- configs[2]: homomorphic add, scalar multiplication (both branches), obfuscate ----------------------------
    ops, ops_ok = None, True
    if not args.no_ops:
        ops = {}
        c2 = be.roll(c)
        out = be.empty(B, s2)
        idx = strided(B, args.ops_sample)
        warm_rows = min(B, 16384)                                  # above the small-batch threshold: warms the kernels that are timed

        def run_op(name, fn, reps, check, note=None):
            fn(warm_rows)                                       # first-launch costs (module load, scratch) stay outside
            barrier()
            evs = be.events(reps)
            t0 = time.perf_counter()
            for a, b in evs:
                be.record(a)
                fn(B)
                be.record(b)
            barrier()
            dt = max_over_ranks([time.perf_counter() - t0])[0]
            if args.inject_fault == name and rank == 0:         # test hook: a wrong row in the result must fail the run
                be.sync()
                out[idx[-1], 0] ^= 1
            ok = check() if rank == 0 else None
            ops[name] = {"value": job_rows * reps / dt, "unit": "ops/s", "reps": reps, "ms_per_pass": dt / reps * 1e3,
                         "launch_ms_avg": sum(be.elapsed_ms(a, b) for a, b in evs) / reps,
                         "bit_exact_strided_sample_vs_gmp_oracle": ok, "rows_checked": len(idx) if rank == 0 else None}
            if note:
                ops[name]["note"] = note
            return ok is not False

        ca_s = lambda: be.np(be.take(c, idx))
        ops_ok &= run_op("raw_add", lambda k: be.mulmod(ctx, c, c2, out, k), 20 if be.name == "hip" else 1,
                         lambda: bool(np.array_equal(be.np(be.take(out, idx)),
                                                     orc.add(n_arr, ca_s(), be.np(be.take(c2, idx)), nthreads=cores))))
        if hasattr(be, "montmul") and hasattr(ctx, "montmul_dev"):
            # resident vectors: ONE Montgomery product per addition, the missing powers of R ("debt") settled by one product
            # with a constant when the residues are needed (phe/ciphertext.py).  Timed as the chain c2 + 8 x c: 8 additions
            # + 1 settling product, the result checked as plain residues c2 * c^8 mod n^2.
            chain = 8
            R = 1 << ctx.mont_radix_bits()
            const = be.upload(native.ints_to_limbs([pow(R, chain + 1, n_int * n_int)], s2))
            tmp = be.empty(B, s2)

            def lazy_chain(k):
                src = c2
                for i in range(chain):
                    dst = tmp if i % 2 == 0 else out
                    be.montmul(ctx, src, c, False, dst, k)
                    src = dst
                be.montmul(ctx, src, const, True, tmp if src is out else out, k)

            def check_lazy():
                final = tmp if chain % 2 == 0 else out               # the settle wrote the buffer the chain did not end in
                sc = np.zeros((len(idx), s1), np.uint32)
                sc[:, 0] = chain
                want = orc.add(n_arr, orc.mul(n_arr, ca_s(), sc, nthreads=cores), be.np(be.take(c2, idx)), nthreads=cores)
                return bool(np.array_equal(be.np(be.take(final, idx)), want))
            ops_ok &= run_op("raw_add_resident_chain", lazy_chain, 4, check_lazy,
                             "per pass: 8 additions at one Montgomery product each + 1 settling product; `value` counts passes")
            ops["raw_add_resident_chain"]["additions_per_s"] = ops["raw_add_resident_chain"]["value"] * chain
            del tmp
        if hasattr(be, "pair_mul") and getattr(ctx, "pair_words", lambda: 0)():
            # resident vectors in the engine's own format (the pair form the exponentiation kernels compute in): an addition is
            # ONE half-width pair product, nothing to settle, no word <-> limb conversion between steps.  Timed like the chain
            # above: c2 + 8 x c on rows that are already resident in pair form = 8 pair products + ONE exit to the canonical
            # residue, checked as c2 * c^8 mod n^2 against libgmp.  The conversion INTO the form (once per vector, like the
            # upload) is timed on its own.
            chain, pw = 8, ctx.pair_words()
            pc, pc2, pt1, pt2 = (be.empty(B, pw) for _ in range(4))
            be.to_pair(ctx, c, pc, B)
            be.to_pair(ctx, c2, pc2, B)

            def pair_chain(k):
                src = pc2
                for i in range(chain):
                    dst = pt1 if i % 2 == 0 else pt2
                    be.pair_mul(ctx, src, pc, dst, k)
                    src = dst
                be.from_pair(ctx, src, out, k)

            def check_pair():
                sc = np.zeros((len(idx), s1), np.uint32)
                sc[:, 0] = chain
                want = orc.add(n_arr, orc.mul(n_arr, ca_s(), sc, nthreads=cores), be.np(be.take(c2, idx)), nthreads=cores)
                return bool(np.array_equal(be.np(be.take(out, idx)), want))
            ops_ok &= run_op("raw_add_resident_pair_form", pair_chain, 4, check_pair,
                             "per pass: 8 additions at one pair product each + 1 exit to the canonical residue; `value` counts passes")
            ops["raw_add_resident_pair_form"]["additions_per_s"] = ops["raw_add_resident_pair_form"]["value"] * chain
            def check_conversion():                                  # out of the form again: every row must be the ciphertext it came from
                be.from_pair(ctx, pt1, out, B)
                be.sync()
                return be.equal(out, c)
            ops_ok &= run_op("to_pair_form", lambda k: be.to_pair(ctx, c, pt1, k), 4, check_conversion,
                             "conversion of resident ciphertext rows into the pair form (once per vector); check = the whole batch "
                             "converted back equals the ciphertexts")
            del pc, pc2, pt1, pt2
        if hasattr(be, "encrypt_owner") and getattr(ctx, "owner_encrypt_offered", lambda: False)():
            # raw_encrypt by the holder of the private key: r^n from its CRT halves (half-width numbers), lifted to n^2 —
            # NOT the headline (that is the public-key path above); the whole batch must equal the public path's ciphertexts
            ops_ok &= run_op("raw_encrypt_key_owner", lambda k: be.encrypt_owner(ctx, m, r, out, k), 2,
                             lambda: be.equal(out, c),
                             "same (m, r) as the headline batch; check = every ciphertext equals the public-key path's")
        scal = {}
        for name, bits, seed in (("raw_mul_float56", 56, 77), ("raw_mul_int64", 63, 78)):
            e = be.rand(B, 2, seed + 10 * rank)
            e[:, 1] &= (0x00ffffff if bits == 56 else 0x7fffffff)
            scal[name] = e

            def check(e=e):
                sc = np.zeros((len(idx), s1), np.uint32)
                sc[:, :2] = be.np(be.take(e, idx))
                return bool(np.array_equal(be.np(be.take(out, idx)), orc.mul(n_arr, ca_s(), sc, nthreads=cores)))
            ops_ok &= run_op(name, lambda k, e=e, bits=bits: be.powmod(ctx, c, e, bits, out, k), 3 if be.name == "hip" else 1, check)
        if hasattr(be, "pair_powmod") and hasattr(ctx, "pair_powmod_dev") and getattr(ctx, "pair_words", lambda: 0)():
            # the same 56-bit scalar multiplication on rows that are resident in the pair form, result in the pair form: no
            # conversion in, no exit (include/phe_hip.h phe_hip_pair_powmod_dev).  Checked by leaving the form once, outside
            # the timed passes: every row must equal what the plain raw_mul_float56 gave (already checked against libgmp).
            pw = ctx.pair_words()
            pc, pout = be.empty(B, pw), be.empty(B, pw)
            be.to_pair(ctx, c, pc, B)
            e56 = scal["raw_mul_float56"]
            ref = be.empty(B, s2)
            be.powmod(ctx, c, e56, 56, ref, B)

            def check_pair_mul():
                be.from_pair(ctx, pout, out, B)
                be.sync()
                return be.equal(out, ref)
            ops_ok &= run_op("raw_mul_float56_resident_pair_form", lambda k: be.pair_powmod(ctx, pc, e56, 56, pout, k), 3, check_pair_mul,
                             "rows resident in the pair form, result in the pair form; check = every row, converted back, equals "
                             "raw_mul_float56's (itself sampled against libgmp)")
            del pc, pout, ref
        # 10 % negative scalars: those rows take invert(c, n^2) as the base and n - s as the exponent (phe/paillier.py:745-749);
        # composed as Engine.raw_mul_signed_dev composes it: one simultaneous inversion of the vector, a per-row select, one powmod
        e = scal["raw_mul_float56"]
        neg = be.mask_every(B, 10)
        inv, base = be.empty(B, s2), be.empty(B, s2)

        subset = hasattr(be, "gather_rows") and hasattr(ctx, "gather_rows_dev")
        neg_idx = be.index_every(B, 10) if subset else None
        sub = be.empty((B + 9) // 10, s2) if subset else None

        def neg_mul(k):
            if subset:
                # what Engine._inverted_where does for few negative rows: only THOSE are inverted (gather, the simultaneous
                # inversion of the subset, scatter into a copy of the vector)
                cnt = (k + 9) // 10
                be.gather_rows(ctx, c, neg_idx, sub, cnt)
                be.invert(ctx, sub, inv, cnt)
                be.copy_rows(base, c, k)
                be.scatter_rows(ctx, inv, neg_idx, base, cnt)
            else:
                be.invert(ctx, c, inv, k)
                be.select_rows(ctx, c, inv, neg, base, k)
            be.powmod(ctx, base, e, 56, out, k)

        def check_neg():
            sc = native.limbs_to_ints(np.ascontiguousarray(be.np(be.take(e, idx))))
            sc = [(n_int - v) if (i % 10 == 0 and v) else v for i, v in zip(idx, sc)]      # negative scalar -v is the residue n - v
            got = be.np(be.take(out, idx))
            return bool(np.array_equal(got, orc.mul(n_arr, ca_s(), native.ints_to_limbs(sc, s1), nthreads=cores)))
        ops_ok &= run_op("raw_mul_float56_neg10pct", neg_mul, 2 if be.name == "hip" else 1, check_neg,
                         "the 10 % negative rows gathered, inverted as a batch of their own (3 modmuls per row + 1 host inversion), "
                         "scattered into a copy of the vector, then one powmod over all rows" if subset else
                         "simultaneous inversion of the whole vector (3 modmuls per row + 1 host inversion) + select + powmod")
        ops_ok &= run_op("obfuscate", lambda k: be.obfuscate(ctx, c, r, out, k), 1,
                         lambda: bool(np.array_equal(be.np(be.take(out, idx)),
                                                     orc.obfuscate(n_arr, ca_s(), be.np(be.take(r, idx)), nthreads=cores))))
        del c2, out, inv, base

    # ---- the reference's own benchmark shape: ONE operation at a time (examples/benchmarks.py:12-29) ------------------------
    # device time of a call on 1 / 256 resident rows (HIP events around back-to-back calls): what the scalar API pays per number
    latency, latency_ok = None, True
    if not args.no_ops and be.name == "hip" and rank == 0:
        latency = {"unit": "ms of device time per call, rows resident", "reps": 10}
        for name, fn, rows in (("raw_decrypt_1_row", dec_step, 1), ("raw_decrypt_256_rows", dec_step, 256), ("raw_encrypt_1_row", enc_step, 1)):
            call = (lambda k: be.decrypt(ctx, c, m_back, k)) if fn is dec_step else (lambda k: be.encrypt(ctx, m, r, c, k))
            call(rows)
            be.sync()
            evs = be.events(1)
            be.record(evs[0][0])
            for _ in range(10):
                call(rows)
            be.record(evs[0][1])
            be.sync()
            latency[name + "_ms"] = be.elapsed_ms(*evs[0]) / 10
            latency[name + "_path"] = ctx.last_launch() if hasattr(ctx, "last_launch") else None
        latency_ok = be.equal(m_back, m)                       # rows rewritten by the small calls still equal the batch's
        latency["bit_exact"] = bool(latency_ok)

    # ---- configs[3]: a shard per GPU under a 3072-bit key + ONE all-gather of the ciphertext shards ---------------
    cfg4, cfg4_ok, lib_state = None, True, None
    if not args.no_config4:
        k4 = golden(args.config4_key_bits)
        t1, t2 = args.config4_key_bits // 32, args.config4_key_bits // 16
        ctx4 = be.context(k4["n"], n_limbs=t1)
        # weak (default): the per-GPU shard of BASELINE configs[3]'s 8M job on 8 GPUs — 2^20 rows on every GPU, whatever N
        # (N = 1: that one shard, ~6.5 s at 3072 bits); strong: the 8M job itself (2^23 rows) cut over the ranks present
        total = args.config4_total or ((1 << 23) if strong else world * (1 << 20))
        lo, hi = shard_bounds(total, world, rank)
        rows = hi - lo
        blk = 1 << 16
        lib_gather_wanted = ((args.lib_allgather or world > 1) and not args.no_lib_allgather and be.name == "hip" and total % world == 0) \
            or bool(os.environ.get("PHE_BENCH_TEST_LIB_GATHER_HANG"))
        # what this rank is about to allocate for the leg, said BEFORE it is allocated (a first N > 1 run that dies of memory
        # should say where): operands m, r + ciphertext shard, the gathered vector (twice with the library's gather), tables
        budget = config4_memory_budget(rows, total, t1, t2, use_dist, lib_gather_wanted)
        print("bench.py rank %d/%d configs[3] memory budget: %s = %.2f GB on this GPU (resident from configs[1]: %.2f GB)"
              % (rank, world, ", ".join("%s %.2f GB" % (k, v / 1e9) for k, v in budget.items()), sum(budget.values()) / 1e9,
                 (2 * B * s1 + B * s2 + B * s1) * 4 / 1e9), file=sys.stderr, flush=True)

        def operands(first, count):
            """rows [first, first+count) of the job's (m, r): a function of the row index only (a seeded stream per 2^16-row
            block), so that any rank — and the checker — can regenerate any row"""
            ms, rs = [], []
            for b in range(first // blk, (first + count - 1) // blk + 1):
                mb, rb = be.rand(blk, t1, 400000 + 2 * b), be.rand(blk, t1, 400001 + 2 * b)
                a, z = max(first, b * blk) - b * blk, min(first + count, (b + 1) * blk) - b * blk
                ms.append(mb[a:z])
                rs.append(rb[a:z])
            mm, rr = be.cat(ms), be.cat(rs)
            clamp_operands(mm, rr, t1, k4["n"])
            return mm, rr

        m4, r4 = operands(lo, rows) if rows else (be.empty(0, t1), be.empty(0, t1))
        c4 = be.empty(rows, t2)
        be.encrypt(ctx4, m4, r4, c4, min(rows, 16384))          # first-launch costs (of the kernels timed below) outside the timed region
        barrier()
        t0 = time.perf_counter()
        be.encrypt(ctx4, m4, r4, c4, rows)
        barrier()
        t_enc = time.perf_counter() - t0
        t0 = time.perf_counter()
        full = all_gather_rows(be.as_tensor(c4), total) if use_dist else be.as_tensor(c4)
        barrier()
        t_gather = time.perf_counter() - t0
        t_enc, t_gather = max_over_ranks([t_enc, t_gather])
        # The same exchange issued by the library's own RCCL communicator (include/phe_hip.h phe_hip_allgather_dev) runs LAST, after
        # every other leg and every check of the line (library_allgather_leg below): it is the one path of this file that no box
        # with more than one GPU has run yet, and a hang inside it must not cost the run its line.
        lib_state = dict(c4=c4, full=full, rows=rows, total=total, t2=t2, ctx4=ctx4) if lib_gather_wanted else None
        if rank == 0:
            bounds = [shard_bounds(total, world, k) for k in range(world)]
            idx = sorted(set([0, total - 1] + [b[0] for b in bounds if b[0] < total] + [max(0, b[1] - 1) for b in bounds] +
                             strided(total, args.config4_sample)))
            ms, rs = [], []
            for b in sorted(set(i // blk for i in idx)):        # the operands of the checked rows, one 2^16-row block at a time
                mb, rb = operands(b * blk, min(blk, total - b * blk))
                sel = [i - b * blk for i in idx if i // blk == b]
                ms.append(be.take(mb, sel))
                rs.append(be.take(rb, sel))
            want = orc.encrypt(native.int_to_limbs(k4["n"], t1), be.np(be.cat(ms)), be.np(be.cat(rs)), nthreads=cores)
            if args.inject_fault == "config4":                  # test hook: a wrong row in the gathered vector must fail the run
                be.sync()
                full[idx[-1], 0] ^= 1
            got = full[idx].cpu().numpy().view(np.uint32)
            cfg4_ok = cfg4_ok and bool(np.array_equal(got, want))
            cfg4 = {"workload": "configs[3]: %d-bit key, %d plaintexts sharded over %d GPU(s) (%d per GPU%s), ONE all-gather "
                                "of the ciphertext shards (phe.sharding.all_gather_rows, backend %s)"
                                % (args.config4_key_bits, total, world, rows,
                                   "" if (strong or args.config4_total) else ": weak scaling, the per-GPU shard of the 8M job on 8 GPUs",
                                   be.dist_backend if use_dist else "none: 1 rank"),
                    "total": total, "rows_per_gpu": rows, "scaling": args.scaling, "memory_budget_bytes_rank0": budget,
                    "encrypt": {"seconds": t_enc, "value": total / t_enc, "unit": "encrypts/s"},
                    "all_gather": {"seconds": t_gather, "bytes_received_per_gpu": total * t2 * 4,
                                   "GBps_per_gpu": total * t2 * 4 / t_gather / 1e9 if use_dist else None},
                    "all_gather_by_library_rccl": None,     # filled in by library_allgather_leg
                    "end_to_end_encrypts_per_s": total / (t_enc + t_gather),
                    "bit_exact_boundaries_and_sample_vs_gmp_oracle": cfg4_ok, "rows_checked": len(idx),
                    "geometry": ctx4.info()}
        del full, c4, m4, r4

    # ---- CPU baseline: the libgmp oracle on all host cores, bounded sample, rank 0 at N = 1 only -------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        pq = s1 // 2
        p_arr, q_arr = native.int_to_limbs(key["p"], pq), native.int_to_limbs(key["q"], pq)

        def cpu_timed(fn, target_s):
            """Time fn(count) on a bounded sample: a short probe sizes the sample for ~target_s seconds."""
            probe = min(B, 32 * cores)                     # large enough that thread start-up does not dominate the estimate
            t0 = time.perf_counter()
            fn(probe)
            rate = probe / max(time.perf_counter() - t0, 1e-6)
            count = int(min(B, max(probe, args.cpu_sample or rate * target_s)))
            t0 = time.perf_counter()
            res = fn(count)
            return count, time.perf_counter() - t0, res

        if args.inject_fault == "cpu_baseline":                 # test hook: a wrong ciphertext inside the libgmp sample must fail the run
            be.sync()
            c[min(1, B - 1), 0] ^= 1
        m_h, r_h, c_h = be.np(m), be.np(r), be.np(c)
        ne, t_enc_cpu, ch = cpu_timed(lambda k: orc.encrypt(n_arr, m_h[:k], r_h[:k], nthreads=cores), 12.0)
        cpu_ok = bool(np.array_equal(ch, c_h[:ne]))
        nd, t_dec_cpu, dh = cpu_timed(lambda k: orc.decrypt(n_arr, p_arr, q_arr, c_h[:k], nthreads=cores), 6.0)
        cpu_ok = cpu_ok and bool(np.array_equal(dh, m_h[:nd]))
        cpu = {"value": ne / t_enc_cpu, "unit": "encrypts/s", "cores": cores, "kind": "port",
               "sample": "first %d of the same (m, r) batch through oracle/paillier_oracle.c "
                         "(libgmp %s mpz_powm = what gmpy2.powmod executes), %d threads, %.1f s; "
                         "decrypt: first %d ciphertexts, %.1f s" % (ne, orc.gmp_version, cores, t_enc_cpu, nd, t_dec_cpu),
               "decrypts_per_s": nd / t_dec_cpu, "matches_gpu": cpu_ok}

    # every comparison the line reports feeds the exit code: the full round trip, the strided libgmp sample, each ops leg, the
    # small-call leg, configs[3] and the cpu_baseline leg's rows (the largest libgmp sample of the line)
    ok = roundtrip_ok and sample_ok is not False and ops_ok and cfg4_ok and latency_ok and (cpu is None or cpu["matches_gpu"])
    if rank == 0:
        enc_mac, dec_mac = mac32_counts(args.key_bits)
        peak, sustained, peak_src = valu_peak_mac32()
        info = ctx.info()
        split = info.get("engine_pub") == "split"
        kname = lambda limbs, eng, mode: "k_modexp_%s<%d,%d,%s>" % (("split" if eng == "split" else "uniform",) + divmod(limbs, 100) + (mode,))
        if "lane_limbs_pub" in info:
            enc_kernel = kname(info["lane_limbs_pub"], info["engine_pub"], "encrypt")
            dec_kernel = kname(info["lane_limbs_priv"], info["engine_priv"], "half_decrypt")
            model_enc, model_dec = executed_mads(args.key_bits, info)
        else:
            enc_kernel, dec_kernel, model_enc, model_dec = "emulator", "emulator", None, None
        counted, counted_src = counted_mads(args.key_bits, info)
        enc_exec = counted["encrypt"] if counted else model_enc
        dec_exec = counted["decrypt"] if counted else model_dec
        if counted and dec_geom and dec_geom != info.get("lane_limbs_priv"):
            by_geom = counted.get("decrypt_by_halves_geometry", {})
            if str(dec_geom) in by_geom:
                dec_exec = by_geom[str(dec_geom)]
                dec_kernel = kname(dec_geom, info["engine_priv"], "half_decrypt")
            else:
                dec_exec = None  # no exact count for the rung that ran: rather no fraction than one for another kernel
        exec_src = ("exact: wave::mad64 calls counted by the CPU wave emulator, %s" % counted_src) if counted else \
            "model (bench.py:executed_mads): no exact count committed for this geometry"
        traffic_unit, traffic_src = measured_traffic_per_unit(enc_kernel) if args.key_bits == 2048 else (None, None)
        valu_unit, valu_src = pmc_valu_per_unit(enc_kernel) if args.key_bits == 2048 else (None, None)
        enc_kernel_s = sum(enc_launch_ms) / len(enc_launch_ms) * 1e-3
        dec_kernel_s = sum(dec_launch_ms) / len(dec_launch_ms) * 1e-3
        value = job_rows * args.steps / enc_dt
        rate = lambda per_unit, seconds: per_unit * B / seconds if per_unit else None
        frac = lambda per_unit, seconds: per_unit * B / seconds / peak if per_unit else None
        roofline = {
            "bound": "valu_int32", "kernel": enc_kernel + " (radix 2^29)",
            "achieved": rate(enc_exec, enc_kernel_s) / 1e12 if enc_exec else None, "peak": peak / 1e12, "unit": "Tmad/s (v_mad_u64_u32 lane-operations = MAC32)",
            "frac": frac(enc_exec, enc_kernel_s),
            "frac_note": "multiply-adds the kernel EXECUTES per second / nominal integer-VALU peak (<= 1 by construction)",
            "kernel_note": "where the key width offers it (>= 2048 bits) r^n runs modulo the scaled modulus n' = k*n = -1 mod 2^29 "
                           "(template flag `unit`: no v_mul_lo per quotient digit) and ONE k_mulmod_staged pass brings the residue to "
                           "(1 + n*m) * r^n mod n^2; launch_ms_avg and the executed count cover both kernels",
            "executed_mad_per_encrypt": enc_exec, "executed_source": exec_src,
            "frac_of_sustained_mad_rate": (rate(enc_exec, enc_kernel_s) / sustained) if (sustained and enc_exec) else None,
            "canonical_mac32_per_encrypt": enc_mac, "canonical_achieved": enc_mac * B / enc_kernel_s / 1e12,
            "canonical_frac": enc_mac * B / enc_kernel_s / peak,
            "canonical_note": "SURVEY.md 8(d) algorithmic MAC32 (full-width Montgomery) x batch / launch time / peak: exceeds "
                              "`frac` because the split-modulus kernels execute fewer multiply-adds than the canonical algorithm",
            "pmc_valu_lane_ops_per_encrypt": valu_unit * 64 if valu_unit else None,
            "mad_share_of_valu_instructions": (enc_exec / (valu_unit * 64)) if (valu_unit and enc_exec) else None,
            "pmc_source": valu_src,
            # committed measurements (rocprofv3 / the emulator count cannot run inside this script): stale = made from other
            # device sources than this tree's (tools/csrc_hash.py) — the figure then describes an OLDER kernel
            "stale": bool(committed_count_is_stale(counted_src) or committed_count_is_stale(traffic_src) or committed_count_is_stale(valu_src)),
            "stale_sources": [f for f in sorted({counted_src, traffic_src, valu_src} - {None}) if committed_count_is_stale(f)],
            "csrc_sha256": current_csrc_hash(),
            "traffic": (traffic_unit * B) if traffic_unit else None,
            "traffic_note": ("HBM+MALL bytes per launch = %.0f B/encrypt (PMC FETCH_SIZE x2 + WRITE_SIZE, %s) x batch; "
                             "algorithmic bytes are %d B/encrypt" % (traffic_unit, traffic_src, (2 * s1 + s2) * 4))
            if traffic_unit else "no PMC run committed for this kernel",
            "launch_ms_avg": enc_kernel_s * 1e3,
            "peak_source": "256 CUs x 4 SIMD x 64 lanes x 2.4 GHz / 4 cycles per v_mad_u64_u32 (half-rate, calibrated)",
            "peak_sustained_microbench": (sustained / 1e12) if sustained else None, "microbench": peak_src,
            "hbm_algorithmic_GBps": (2 * s1 + s2) * 4 * B / enc_kernel_s / 1e9, "hbm_peak_GBps": 8000.0,
            "decrypt": {"kernel": "2 x %s + k_decrypt_tail_wave" % dec_kernel,
                        "achieved": rate(dec_exec, dec_kernel_s) / 1e12 if dec_exec else None, "frac": frac(dec_exec, dec_kernel_s),
                        "executed_mad_per_decrypt": dec_exec, "canonical_mac32_per_decrypt": dec_mac,
                        "canonical_frac": dec_mac * B / dec_kernel_s / peak, "launch_ms_avg": dec_kernel_s * 1e3},
        }
        if ops:
            canon = mac32_ops(args.key_bits)
            exec_key = {"raw_add": "raw_add", "raw_mul_float56": "raw_mul_56bit", "raw_mul_int64": "raw_mul_63bit",
                        "obfuscate": "obfuscate", "raw_encrypt_key_owner": "encrypt_key_owner"}
            canon["raw_encrypt_key_owner"] = enc_mac
            for name, rec in ops.items():
                per_gpu = rec["value"] / world
                if name in ("raw_add_resident_pair_form", "to_pair_form"):
                    # executed multiply-adds, exact: a pair product is 5 H^2 (pair_pass3), the exit 8 H^2 (montmul_q 2, montmac2 3,
                    # montmul 2, mul_wide 1), the conversion in 13 H^2 (two chunk sweeps of 4 and one pair product), H = limbs of n
                    Hn = ctx.pair_words() // 2
                    if name == "to_pair_form":
                        ex = 13 * Hn * Hn
                        rec["roofline"] = {"bound": "valu_int32", "executed_mad_per_op": ex, "frac": ex * per_gpu / peak}
                    else:
                        adds = rec["additions_per_s"] / world
                        ex_pass = (8 * 5 + 8) * Hn * Hn
                        rec["roofline"] = {"bound": "valu_int32", "canonical_mac32_per_op": canon["raw_add"],
                                           "executed_mad_per_addition": ex_pass / 8, "frac": ex_pass / 8 * adds / peak,
                                           "hbm_algorithmic_GBps": (3 * ctx.pair_words() * 4 * 8 + (ctx.pair_words() + s2) * 4) / 8 * adds / 1e9}
                    continue
                if name == "raw_add_resident_chain":
                    adds = rec["additions_per_s"] / world
                    two = counted.get("raw_add_two_montgomery_products", counted.get("raw_add")) if counted else None
                    half = (two / 2) if two else None                           # one of the two products of the Montgomery form
                    rec["roofline"] = {"bound": "valu_int32", "canonical_mac32_per_op": canon["raw_add"],
                                       "canonical_frac": canon["raw_add"] * adds / peak,
                                       "executed_mad_per_op": half * 9 / 8 if half else None,
                                       "frac": (half * 9 / 8 * adds / peak) if half else None,
                                       "hbm_algorithmic_GBps": 3 * s2 * 4 * 9 / 8 * adds / 1e9}
                    continue
                ck = "raw_mul_float56" if name.startswith("raw_mul_float56") else name
                ex = counted.get(exec_key[ck]) if counted else None
                rec["roofline"] = {"bound": "valu_int32", "canonical_mac32_per_op": canon[ck],
                                   "canonical_frac": canon[ck] * per_gpu / peak,
                                   "executed_mad_per_op": ex, "frac": (ex * per_gpu / peak) if ex else None}
            # HBM traffic and VALU instructions of each op's kernel as the PMC counters saw them (separate rocprofv3 --pmc passes on the
            # committed tree: tools/gpu_pmc_traffic.sh), per launch like `traffic` of the headline roofline
            pmc_ops, pmc_ops_src = measured_ops_traffic(args.key_bits)
            for name, rec in ops.items():
                t = pmc_ops.get(name)
                if not t or "roofline" not in rec:
                    continue
                rec["roofline"].update(
                    traffic=t.get("bytes_per_row") and t["bytes_per_row"] * B, traffic_bytes_per_op=t.get("bytes_per_row"),
                    pmc_valu_wave_instructions_per_op=t.get("valu_wave_instructions_per_row"), pmc_kernel=t.get("kernel"),
                    pmc_source=pmc_ops_src, pmc_stale=committed_count_is_stale(pmc_ops_src))
            add_bytes = 3 * s2 * 4
            if counted and counted.get("raw_add_form"):
                ops["raw_add"]["roofline"]["form"] = counted["raw_add_form"]
            ops["raw_add"]["roofline"].update(hbm_algorithmic_GBps=add_bytes * ops["raw_add"]["value"] / world / 1e9,
                                              hbm_frac=add_bytes * ops["raw_add"]["value"] / world / 8e12)
        out = {
            "metric": METRIC, "value": value, "unit": "encrypts/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": enc_dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "u32 (29-bit limbs, v_mad_u64_u32 with 64-bit accumulate)",
            "data": "synthetic" if be.name == "hip" else "synthetic, SELFTEST on the CPU wave emulator — not a measurement",
            "rccl_ranks": dist.get_world_size() if use_dist else 1, "backend": be.dist_backend if use_dist else "single process",
            "config": {"workload": "configs[1]: %d-bit key, %s, raw_encrypt then raw_decrypt, operands resident in HBM"
                                   % (args.key_bits, ("%d-plaintext job cut into contiguous shards (strong scaling: %d on rank 0)" % (job_rows, B))
                                      if strong else "%d-plaintext batch per GPU" % B),
                       "key_bits": args.key_bits, "batch_per_gpu": B, "batch_whole_job": job_rows, "parallelism": "batch-sharded x%d" % world,
                       "geometry": info},
            "decrypt": {"value": job_rows * args.steps / dec_dt, "unit": "decrypts/s",
                        "ms_per_step": dec_dt / args.steps * 1e3},
            "bit_exact": {"roundtrip_full_batch": roundtrip_ok, "strided_sample_vs_gmp_oracle": sample_ok,
                          "strided_sample_rows": sample_rows},
            "roofline": roofline,
            "ops": ops, "latency": latency, "config4": cfg4,
            "cpu_baseline": cpu,
        }
        if cpu:
            out["speedup_vs_cpu_all_cores"] = {"encrypt": value / cpu["value"],
                                               "decrypt": out["decrypt"]["value"] / cpu["decrypts_per_s"]}
    else:
        out = None
    if use_dist:
        ok = max_over_ranks([0.0 if ok else 1.0])[0] == 0.0

    def print_line():
        if rank == 0:
            sys.stdout.flush()
            os.write(json_fd, (json.dumps(out) + "\n").encode())

    if lib_state is not None:
        ok = library_allgather_leg(args, be, native, dist, use_dist, rank, world, lib_state, out, ok, print_line, barrier, max_over_ranks)
        lib_state = None
    print_line()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        sys.exit(1)


def library_allgather_leg(args, be, native, dist, use_dist, rank, world, st, out, ok, print_line, barrier, max_over_ranks):
    """configs[3]'s all-gather once more, issued by the library's own RCCL communicator (include/phe_hip.h phe_hip_comm_create /
    phe_hip_allgather_dev: the path of a host without a process group; the id travels over the torch group here), bits compared
    with torch's gather.  Runs after every other leg, with the line assembled and the ranks' verdicts exchanged: if the second
    RCCL instance of the process hangs (communicator set-up or the collective itself), a watchdog prints the line with
    `all_gather_by_library_rccl.error` and ends the process with the exit code the other legs earned.  Returns ok."""
    import threading
    from phe.sharding import library_communicator, shard_bounds
    c4, full, rows, total, t2, ctx4 = st["c4"], st["full"], st["rows"], st["total"], st["t2"], st["ctx4"]

    def record(result):
        if rank == 0 and out is not None and out.get("config4"):
            out["config4"]["all_gather_by_library_rccl"] = result

    def on_timeout():
        record({"error": "no answer within %.0f s (--lib-allgather-timeout): communicator set-up or the collective hangs; "
                         "the rest of the line was measured before this leg" % args.lib_allgather_timeout})
        print("bench.py rank %d: the library's RCCL all-gather did not finish in %.0f s: the line goes out without it"
              % (rank, args.lib_allgather_timeout), file=sys.stderr, flush=True)
        print_line()
        os._exit(0 if ok else 1)     # (the main thread sits in a collective that will not return: no clean way out)

    dog = threading.Timer(args.lib_allgather_timeout, on_timeout)
    dog.daemon = True
    dog.start()
    if os.environ.get("PHE_BENCH_TEST_LIB_GATHER_HANG"):        # test hook (tests/test_bench_contract.py): the leg never returns
        time.sleep(10 * args.lib_allgather_timeout + 60)

    def exchange(uid):
        box = [uid]
        if use_dist:
            dist.broadcast_object_list(box, src=0)
        return box[0]
    # A communicator that cannot be created (no librccl to dlopen, an RCCL error) must not take the headline with it:
    # every rank learns whether ALL ranks got one (a rank that went on alone would hang the others in the collective),
    # the line says what happened, and only a gather that RAN and differs from torch's makes the run exit non-zero.
    comm, lib_error, uid, lib_gather = None, None, None, None
    if rank == 0:                     # (the id first, on its own: a rank 0 that fails here must still reach the broadcast)
        try:
            uid = native.comm_unique_id()
        except Exception as e:  # noqa: BLE001
            lib_error = "%s: %s" % (type(e).__name__, e)
    uid = exchange(uid)
    if uid is not None:
        try:
            comm = library_communicator(ctx4, rank, world, lambda _ignored: uid)
        except Exception as e:  # noqa: BLE001
            lib_error = "%s: %s" % (type(e).__name__, e)
    if comm is None:
        print("bench.py rank %d: the library's RCCL communicator could not be created: %s" % (rank, lib_error or "no id from rank 0"),
              file=sys.stderr, flush=True)
    if max_over_ranks([0.0 if comm is not None else 1.0])[0] != 0.0:
        if comm is not None:
            comm.close()
        comm = None
        lib_gather = {"error": lib_error or "another rank could not create its communicator"}
    if comm is not None:
        full2 = be.empty(total, t2)
        comm.allgather_dev(c4.data_ptr(), full2.data_ptr(), rows, t2, be.stream)      # first call: connection set-up
        barrier()
        t0 = time.perf_counter()
        comm.allgather_dev(c4.data_ptr(), full2.data_ptr(), rows, t2, be.stream)
        barrier()
        t_lib = max_over_ranks([time.perf_counter() - t0])[0]
        lib_gather = {"seconds": t_lib, "GBps_per_gpu": total * t2 * 4 / t_lib / 1e9,
                      "same_bits_as_torch_all_gather": be.equal(full2, full)}
        if args.inject_fault == "lib_allgather":               # test hook: a library gather that differs must fail the run
            lib_gather["same_bits_as_torch_all_gather"] = False
        if not lib_gather["same_bits_as_torch_all_gather"]:
            # say WHERE: the first row on which the library's gather and torch's differ (the run exits non-zero)
            diff = (be.as_tensor(full2) != be.as_tensor(full)).any(dim=1).nonzero()
            first = int(diff[0]) if len(diff) else -1
            lib_gather["first_differing_row"] = first
            lib_gather["first_differing_row_owner_rank"] = next((k for k in range(world) if shard_bounds(total, world, k)[0] <= first < shard_bounds(total, world, k)[1]), None)
            print("bench.py rank %d: library RCCL all-gather differs from torch's at row %d of %d (%d rows differ)"
                  % (rank, first, total, len(diff)), file=sys.stderr, flush=True)
        same = max_over_ranks([0.0 if lib_gather["same_bits_as_torch_all_gather"] else 1.0])[0] == 0.0   # every rank holds both vectors
        if rank == 0 and out is not None and out.get("config4"):
            out["config4"]["bit_exact_boundaries_and_sample_vs_gmp_oracle"] = bool(out["config4"]["bit_exact_boundaries_and_sample_vs_gmp_oracle"] and same)
        ok = ok and same
        comm.close()
        del full2
    dog.cancel()
    record(lib_gather)
    return ok


if __name__ == "__main__":
    main()
