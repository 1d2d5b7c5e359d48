#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on the config it is quoted on.

Metric : "Paillier 2048-bit encrypts/sec + decrypts/sec per node (bit-exact)"
Config : configs[1] — 2048-bit key, 1M-plaintext batch encrypt + decrypt on 1 MI355X (per GPU; weak scaling).

A "step" is one pass of raw_encrypt over the batch (`value` = encrypts/s, the number the >=10x target is
stated in); the same K steps of raw_decrypt are timed as a second region and reported under "decrypt".
Inputs (plaintexts m, obfuscators r, and for decrypt the ciphertexts) are resident in HBM before the
timed region starts.  One process per GPU; N>1 is launched by torch.distributed.run and ranks only meet
at the barriers (the batch is embarrassingly sharded: no data-path collective).

Also printed in the same JSON line:
  roofline     — dominant kernel (k_modexp_split<4,18,encrypt> at 2048 bits): algorithmic MAC32 (SURVEY.md 8(d):
                 the canonical full-width Montgomery count) per launch / average launch duration measured with
                 HIP events on the launch stream, against the integer-VALU peak calibrated by csrc/microbench.hip
                 (profiles/microbench_*.json).  The split-modulus kernels execute fewer multiply-adds than that
                 canonical count (csrc/split_core.h), so `frac` can exceed 1; `executed` gives the multiply-adds
                 the kernel really issues and their fraction of the same peak.
  cpu_baseline — the libgmp oracle (what gmpy2 executes) on all host cores over a bounded sample of the same
                 workload, rank 0 at N=1 only.
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "python-paillier_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

METRIC = "Paillier 2048-bit encrypts/sec + decrypts/sec per node (bit-exact)"


def mac32_counts(key_bits):
    """Algorithmic multiply-accumulates per op, exactly SURVEY.md 8(d)."""
    s2, s1, sh = key_bits // 16, key_bits // 32, key_bits // 64
    mont = lambda s: 2 * s * s + s
    E = lambda t: t + -(-t // 6) + 16
    enc = s1 * s1 + (E(key_bits) + 3) * mont(s2)
    dec = 2 * mont(s1) + 2 * (E(key_bits // 2) + 2) * mont(s1) + 3 * sh * sh + 4 * mont(sh)
    return enc, dec


def executed_mads(key_bits, info):
    """v_mad_u64_u32 lane-operations one encrypt / one decrypt really executes (29-bit limbs; schedule counted like
    SURVEY's E(t): t squarings, ceil(t/(w+1)) + 2^(w-1) products for the window w in use).  Split engine: a squaring is 4 H^2, a product 5 H^2, entry
    4 H^2 per input chunk (+5 H^2 when more than one), exit ~10 H^2 (csrc/split_core.h); full-width engine: 2 S^2 per
    Montgomery product."""
    E_sq = lambda t: t
    # sliding windows of w = 6 bits (32 odd powers) from ~1000-bit exponents on, w = 5 (16) below: key_setup.h:pick_window
    E_mul = lambda t: (-(-t // 7) + 32) if t > 900 else (-(-t // 6) + 16)

    def modexp(t, lane_limbs, split, in_bits, extra):
        G, L = divmod(lane_limbs, 100)
        H = G * L
        if split:
            chunks = max(1, -(-in_bits // (29 * H)))
            return (4 * E_sq(t) + 5 * E_mul(t) + 4 * chunks + (5 if chunks > 1 else 0) + 10 + extra) * H * H
        return (E_sq(t) + E_mul(t) + 3) * 2 * H * H

    enc = modexp(key_bits, info["lane_limbs_pub"], info["engine_pub"] == "split", key_bits, 2)
    dec = 2 * modexp(key_bits // 2, info["lane_limbs_priv"], info["engine_priv"] == "split", 2 * key_bits, 0)
    return enc, dec


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
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            t = json.load(f)[kernel_key]
        return (t["fetch_bytes"] + t["write_bytes"]) / float(t["batch"]), os.path.basename(files[-1])
    except Exception:
        return None, None


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1 << 20, help="plaintexts per GPU per step")
    ap.add_argument("--key-bits", type=int, default=2048, choices=[1024, 2048, 3072])
    ap.add_argument("--cpu-sample", type=int, default=0, help="elements for the CPU baseline (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from phe import _native as native

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    use_dist = world > 1 or "RANK" in os.environ          # launched by torch.distributed.run (also with 1 rank)
    torch.cuda.set_device(local_rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)

    with open(os.path.join(ROOT, "tests", "golden", "paillier_%d.json" % args.key_bits)) as f:
        g = json.load(f)
    H = lambda k: int(g[k], 16)
    n_int = H("n")
    s1, s2 = args.key_bits // 32, args.key_bits // 16
    ctx = native.Context(n_int, H("p"), H("q"), H("hp"), H("hq"), H("p_inverse"), device=local_rank, n_limbs=s1)
    if args.blocks_per_cu:
        ctx.set_blocks_per_cu(args.blocks_per_cu)

    # ---- synthetic inputs, generated on the device (resident in HBM before any timed region) ----
    B = args.batch
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    m = torch.randint(-2 ** 31, 2 ** 31, (B, s1), dtype=torch.int32, device=dev, generator=gen)
    r = torch.randint(-2 ** 31, 2 ** 31, (B, s1), dtype=torch.int32, device=dev, generator=gen)
    m[:, s1 - 1] = 0                      # m < n (n has exactly key_bits bits)
    r[:, s1 - 1] &= 0x3fffffff            # r < n
    r[:, 0] |= 1                          # r != 0
    c = torch.empty((B, s2), dtype=torch.int32, device=dev)
    m_back = torch.empty((B, s1), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def enc_step():
        ctx.encrypt_dev(m.data_ptr(), r.data_ptr(), c.data_ptr(), B, stream)

    def dec_step():
        ctx.decrypt_dev(c.data_ptr(), m_back.data_ptr(), B, stream)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    def timed(step_fn, steps):
        """K steps bracketed by barrier + synchronize on both sides; also per-launch HIP-event durations."""
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        barrier()
        t0 = time.perf_counter()
        for a, b in evs:
            a.record()
            step_fn()
            b.record()
        barrier()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        launch_ms = [a.elapsed_time(b) for a, b in evs]
        return dt, launch_ms

    for _ in range(args.warmup):
        enc_step()
    enc_dt, enc_launch_ms = timed(enc_step, args.steps)
    for _ in range(args.warmup):
        dec_step()
    dec_dt, dec_launch_ms = timed(dec_step, args.steps)

    # ---- bit-exactness of what was just timed -------------------------------------------------
    roundtrip_ok = bool(torch.equal(m_back, m))
    sample_ok = None
    cpu = None
    if rank == 0:
        from oracle.paillier_oracle import COracle
        orc = COracle()
        idx = torch.arange(0, B, max(1, B // 64), device=dev)[:64]
        to_np = lambda t: t.cpu().numpy().view(np.uint32)
        n_arr = native.int_to_limbs(n_int, s1)
        want = orc.encrypt(n_arr, to_np(m[idx]), to_np(r[idx]), nthreads=host_cores())
        sample_ok = bool(np.array_equal(to_np(c[idx]), want))
        if world == 1 and not args.no_cpu_baseline:
            cores = host_cores()
            pq = s1 // 2
            p_arr, q_arr = native.int_to_limbs(H("p"), pq), native.int_to_limbs(H("q"), pq)

            def cpu_timed(fn, target_s):
                """Time fn(count) on a bounded sample: a short probe sizes the sample for ~target_s seconds."""
                probe = min(B, max(cores, 2 * cores))
                t0 = time.perf_counter()
                fn(probe)
                rate = probe / max(time.perf_counter() - t0, 1e-6)
                count = int(min(B, max(probe, args.cpu_sample or rate * target_s)))
                t0 = time.perf_counter()
                res = fn(count)
                return count, time.perf_counter() - t0, res

            ne, t_enc, ch = cpu_timed(lambda k: orc.encrypt(n_arr, to_np(m[:k]), to_np(r[:k]), nthreads=cores), 12.0)
            cpu_ok = bool(np.array_equal(ch, to_np(c[:ne])))
            nd, t_dec, dh = cpu_timed(lambda k: orc.decrypt(n_arr, p_arr, q_arr, to_np(c[:k]), nthreads=cores), 6.0)
            cpu_ok = cpu_ok and bool(np.array_equal(dh, to_np(m[:nd])))
            cpu = {"value": ne / t_enc, "unit": "encrypts/s", "cores": cores, "kind": "port",
                   "sample": "first %d of the same (m, r) batch through oracle/paillier_oracle.c "
                             "(libgmp %s mpz_powm = what gmpy2.powmod executes), %d threads, %.1f s; "
                             "decrypt: first %d ciphertexts, %.1f s" % (ne, orc.gmp_version, cores, t_enc, nd, t_dec),
                   "decrypts_per_s": nd / t_dec, "matches_gpu": cpu_ok}

    if rank == 0:
        enc_mac, dec_mac = mac32_counts(args.key_bits)
        peak, sustained, peak_src = valu_peak_mac32()
        info = ctx.info()
        kname = lambda limbs, eng, mode: "k_modexp_%s<%d,%d,%s>" % (("split" if eng == "split" else "uniform",) + divmod(limbs, 100) + (mode,))
        enc_kernel = kname(info["lane_limbs_pub"], info["engine_pub"], "encrypt")
        dec_kernel = kname(info["lane_limbs_priv"], info["engine_priv"], "half_decrypt")
        traffic_unit, traffic_src = measured_traffic_per_unit(enc_kernel) if args.key_bits == 2048 else (None, None)
        enc_exec, dec_exec = executed_mads(args.key_bits, info)
        enc_kernel_s = sum(enc_launch_ms) / len(enc_launch_ms) * 1e-3
        dec_kernel_s = sum(dec_launch_ms) / len(dec_launch_ms) * 1e-3
        achieved = enc_mac * B / enc_kernel_s
        value = world * B * args.steps / enc_dt
        out = {
            "metric": METRIC, "value": value, "unit": "encrypts/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": enc_dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32 (29-bit limbs, v_mad_u64_u32 with 64-bit accumulate)",
            "data": "synthetic",
            "config": {"workload": "configs[1]: %d-bit key, %d-plaintext batch per GPU, raw_encrypt then raw_decrypt, "
                                   "operands resident in HBM" % (args.key_bits, B),
                       "key_bits": args.key_bits, "batch_per_gpu": B, "parallelism": "batch-sharded x%d" % world,
                       "geometry": info},
            "decrypt": {"value": world * B * args.steps / dec_dt, "unit": "decrypts/s",
                        "ms_per_step": dec_dt / args.steps * 1e3},
            "bit_exact": {"roundtrip_full_batch": roundtrip_ok, "strided_sample_vs_gmp_oracle": sample_ok},
            "roofline": {
                "bound": "valu_int32", "kernel": enc_kernel + " (radix 2^29)",
                "achieved": achieved / 1e12, "peak": peak / 1e12, "unit": "TMAC32/s", "frac": achieved / peak,
                "achieved_note": "SURVEY.md 8(d) canonical MAC32 per encrypt (full-width Montgomery) x batch / launch time",
                "executed": {"mad_per_encrypt": enc_exec, "rate_Tmad_per_s": enc_exec * B / enc_kernel_s / 1e12,
                             "frac_of_peak": enc_exec * B / enc_kernel_s / peak,
                             "note": "v_mad_u64_u32 lane-operations the kernel really issues (29-bit limbs)"},
                "traffic": (traffic_unit * B) if traffic_unit else None,
                "traffic_note": ("HBM+MALL bytes per launch = %.0f B/encrypt (PMC FETCH_SIZE x2 + WRITE_SIZE, %s) x batch; "
                                 "algorithmic bytes are %d B/encrypt" % (traffic_unit, traffic_src, (2 * s1 + s2) * 4))
                if traffic_unit else "no PMC run committed for this kernel",
                "mac32_per_encrypt": enc_mac, "launch_ms_avg": enc_kernel_s * 1e3,
                "peak_source": "256 CUs x 4 SIMD x 64 lanes x 2.4 GHz / 4 cycles per v_mad_u64_u32 (half-rate, calibrated)",
                "peak_sustained_microbench": (sustained / 1e12) if sustained else None, "microbench": peak_src,
                "hbm_algorithmic_GBps": (2 * s1 + s2) * 4 * B / enc_kernel_s / 1e9, "hbm_peak_GBps": 8000.0,
                "decrypt": {"kernel": "2 x %s + k_decrypt_tail" % dec_kernel,
                            "achieved": dec_mac * B / dec_kernel_s / 1e12, "frac": dec_mac * B / dec_kernel_s / peak,
                            "mac32_per_decrypt": dec_mac, "launch_ms_avg": dec_kernel_s * 1e3,
                            "executed": {"mad_per_decrypt": dec_exec, "rate_Tmad_per_s": dec_exec * B / dec_kernel_s / 1e12,
                                         "frac_of_peak": dec_exec * B / dec_kernel_s / peak}},
            },
            "cpu_baseline": cpu,
        }
        if cpu:
            out["speedup_vs_cpu_all_cores"] = {"encrypt": value / cpu["value"],
                                               "decrypt": out["decrypt"]["value"] / cpu["decrypts_per_s"]}
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if not roundtrip_ok or sample_ok is False:
        sys.exit(1)


if __name__ == "__main__":
    main()
