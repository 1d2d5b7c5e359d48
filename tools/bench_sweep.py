#!/usr/bin/env python3
"""Batch-size sweep of the hot path on resident operands: encrypt / decrypt / _raw_add / _raw_mul
(56-bit scalars) / additions in the pair form, batches 2^10 ... 2^20, one key size.  This is the shape of the reference's
own benchmark (examples/benchmarks.py:38-71: a fixed number of operations per key size) taken across batch sizes: which
rung of the geometry ladder each size takes, its rate, and the rate as a fraction of the 2^20 rate.

A point is `reps` launches queued back to back on one stream, one synchronisation at the end (HIP events around the
region): the steady rate of a stream of batches of that size, launch gaps included.  Every size is checked: full round
trip decrypt(encrypt(m)) = m on the device, a strided sample of every op against the libgmp oracle.

  python tools/bench_sweep.py [--key-bits 2048] [--min 10] [--max 20] [--group G]  > profiles/rNN_batch_sweep.json
(--group G pins every size to one rung)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "python-paillier_amd")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--key-bits", type=int, default=2048)
    ap.add_argument("--min", type=int, default=10)
    ap.add_argument("--max", type=int, default=20)
    ap.add_argument("--group", type=int, default=0)
    ap.add_argument("--budget-ms", type=float, default=400.0, help="device time spent per point (sets the repetitions)")
    ap.add_argument("--ops", default="encrypt,decrypt,add,mul,pair_add")
    ap.add_argument("--table", action="store_true", help="print a text table to stderr as well")
    args = ap.parse_args()
    import numpy as np
    import torch
    from phe import _native as native
    from oracle.paillier_oracle import COracle
    dev = torch.device("cuda", 0)
    gdir = os.path.join(ROOT, "tests", "golden")
    if os.path.exists(os.path.join(gdir, "paillier_%d.json" % args.key_bits)):
        g = json.load(open(os.path.join(gdir, "paillier_%d.json" % args.key_bits)))
    else:                                                  # key sizes with committed primes only: the key constants from them
        g = json.load(open(os.path.join(gdir, "paillier_%d_primes.json" % args.key_bits)))
        p_, q_ = sorted((int(g["p"], 16), int(g["q"], 16)))
        n_ = p_ * q_
        h_ = lambda x: pow((pow(n_ + 1, x - 1, x * x) - 1) // x, -1, x)     # phe/paillier.py:356-360
        g = {k: "%x" % v for k, v in dict(n=n_, p=p_, q=q_, hp=h_(p_), hq=h_(q_), p_inverse=pow(p_, -1, q_)).items()}
    H = lambda k: int(g[k], 16)
    n_int = H("n")
    s1, s2 = args.key_bits // 32, args.key_bits // 16
    ctx = native.Context(n_int, H("p"), H("q"), H("hp"), H("hq"), H("p_inverse"), n_limbs=s1)
    if args.group:
        ctx.set_group(args.group)
    top = 1 << args.max
    gen = torch.Generator(device=dev)
    gen.manual_seed(11)
    rnd = lambda rows, cols: torch.randint(-2 ** 31, 2 ** 31, (rows, cols), dtype=torch.int32, device=dev, generator=gen)
    m, r = rnd(top, s1), rnd(top, s1)
    m[:, s1 - 1] = 0
    r[:, s1 - 1] &= 0x3fffffff
    r[:, 0] |= 1
    e = rnd(top, 2)
    e[:, 1] &= 0x00ffffff
    c = torch.empty((top, s2), dtype=torch.int32, device=dev)
    c2 = torch.empty_like(c)
    out = torch.empty_like(c)
    back = torch.empty((top, s1), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()
    st = stream.cuda_stream
    ctx.encrypt_dev(m.data_ptr(), r.data_ptr(), c.data_ptr(), top, st)
    ctx.encrypt_dev(r.data_ptr(), m.data_ptr(), c2.data_ptr(), top, st)
    pair_words = ctx.pair_words()
    pa = pb = pout = None
    if pair_words:
        pa = torch.empty((top, pair_words), dtype=torch.int32, device=dev)
        pb = torch.empty_like(pa)
        pout = torch.empty_like(pa)
        ctx.to_pair_dev(c.data_ptr(), pa.data_ptr(), top, st)
        ctx.to_pair_dev(c2.data_ptr(), pb.data_ptr(), top, st)
    torch.cuda.synchronize()
    orc = COracle()
    n_arr = native.int_to_limbs(n_int, s1)
    p_arr = native.int_to_limbs(H("p"), s1 // 2)
    q_arr = native.int_to_limbs(H("q"), s1 // 2)
    to_np = lambda t: t.cpu().numpy().view(np.uint32)
    ops = args.ops.split(",")

    def timed(fn, est_ms):
        """average milliseconds of one launch of fn in a back-to-back stream of them (HIP events on the launch stream)"""
        reps = int(max(3, min(2000, args.budget_ms / max(est_ms, 1e-3))))
        fn()
        stream.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(reps):
            fn()
        b.record(stream)
        b.synchronize()
        return a.elapsed_time(b) / reps, reps

    # rough per-row costs (ms) to size the repetitions; refined from the first measurement of each op
    est = {"encrypt": 1.8e-3, "decrypt": 0.5e-3, "add": 4e-6, "mul": 7e-5, "pair_add": 1.5e-6, "pair_add_rowb": 1.5e-6}
    points = []
    for lg in range(args.min, args.max + 1):
        B = 1 << lg
        row = {"log2_batch": lg, "batch": B}
        idx = torch.arange(0, B, max(1, B // 24), device=dev)[:24]
        for op in ops:
            if op.startswith("pair_add") and not pair_words:
                continue
            if op == "encrypt":
                fn = lambda: ctx.encrypt_dev(m.data_ptr(), r.data_ptr(), out.data_ptr(), B, st)
            elif op == "decrypt":
                fn = lambda: ctx.decrypt_dev(c.data_ptr(), back.data_ptr(), B, st)
            elif op == "add":
                fn = lambda: ctx.mulmod_dev(c.data_ptr(), c2.data_ptr(), out.data_ptr(), B, st)
            elif op == "mul":
                fn = lambda: ctx.powmod_dev(c.data_ptr(), e.data_ptr(), 2, 56, out.data_ptr(), B, st)
            elif op == "pair_add_rowb":                          # diagnostic: the second operand is ONE row (a third of the traffic)
                fn = lambda: ctx.pair_mul_dev(pa.data_ptr(), pb.data_ptr(), True, pout.data_ptr(), B, st)
            else:
                fn = lambda: ctx.pair_mul_dev(pa.data_ptr(), pb.data_ptr(), False, pout.data_ptr(), B, st)
            ms, reps = timed(fn, est[op] * B)
            info = ctx.last_launch()
            entry = {"per_s": B / ms * 1e3, "ms": ms, "reps": reps}
            if op in ("encrypt", "mul", "pair_add", "pair_add_rowb", "add"):
                entry["geom"] = info["geom_pub"] if op != "add" else None
            if op == "encrypt":
                entry["scaled_modulus"] = bool(info["path"] & ctx.PATH_UNIT)
                ok = np.array_equal(to_np(out[idx]), orc.encrypt(n_arr, to_np(m[idx]), to_np(r[idx]), nthreads=8))
                ok = ok and bool(torch.equal(out[:B], c[:B]))                      # every row equals the 2^max launch's row
            elif op == "decrypt":
                entry["geom"] = info["geom_priv"]
                entry["halves_side_by_side"] = bool(info["path"] & ctx.PATH_SIDE_BY_SIDE)
                ok = bool(torch.equal(back[:B], m[:B]))                            # full round trip
                ok = ok and np.array_equal(to_np(back[idx]), orc.decrypt(n_arr, p_arr, q_arr, to_np(c[idx]), nthreads=8))
            elif op == "add":
                ok = np.array_equal(to_np(out[idx]), orc.add(n_arr, to_np(c[idx]), to_np(c2[idx]), nthreads=8))
            elif op == "mul":
                sc = np.zeros((len(idx), s1), np.uint32)
                sc[:, :2] = to_np(e[idx])
                ok = np.array_equal(to_np(out[idx]), orc.mul(n_arr, to_np(c[idx]), sc, nthreads=8))
            else:
                ctx.from_pair_dev(pout.data_ptr(), None, out.data_ptr(), B, st)
                torch.cuda.synchronize()
                other = to_np(c2[idx]) if op == "pair_add" else np.repeat(to_np(c2[:1]), len(idx), axis=0)
                ok = np.array_equal(to_np(out[idx]), orc.add(n_arr, to_np(c[idx]), other, nthreads=8))
            entry["bit_exact"] = bool(ok)
            row[op] = entry
        points.append(row)
    last = points[-1]
    for row in points:
        for op in ops:
            if op in row and op in last:
                row[op]["frac_of_largest"] = row[op]["per_s"] / last[op]["per_s"]
    pub, priv = ctx.ladder()
    res = {"key_bits": args.key_bits, "ladder_pub": pub, "ladder_priv": priv, "forced_group": args.group,
           "timing": "HIP events around `reps` back-to-back launches, one stream",
           "device": torch.cuda.get_device_name(0), "points": points,
           "all_bit_exact": all(row[op]["bit_exact"] for row in points for op in ops if op in row)}
    worst = {}
    for op in ops:
        vals = [(row[op]["frac_of_largest"], row["log2_batch"]) for row in points if op in row and row["log2_batch"] >= 13]
        if vals:
            worst[op] = {"min_frac_from_2^13": min(vals)[0], "at_log2_batch": min(vals)[1]}
    res["worst_point_from_2^13"] = worst
    print(json.dumps(res))
    if args.table:
        w = sys.stderr.write
        w("log2B " + "".join("%28s" % op for op in ops) + "\n")
        for row in points:
            w("%5d " % row["log2_batch"] + "".join(
                ("%12.4g/s %4d%% g%-5s" % (row[op]["per_s"], round(100 * row[op]["frac_of_largest"]), row[op].get("geom"))).rjust(28)
                if op in row else " " * 28 for op in ops) + "\n")


if __name__ == "__main__":
    main()
