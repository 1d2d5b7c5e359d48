#!/usr/bin/env python3
"""The measured geometry ladder: launch times of every rung, pinned in turn, per key width, kernel family
and batch size — the table the library picks rungs from (include/phe_hip.h phe_hip_ctx_load_ladder) instead of the hand-fitted
estimate `rows * (19 L + 52) * max(1, waves per SIMD)` with its measured-at-3072-bits factors (csrc/phe_hip.hip rung_cost).

    python tools/calibrate_ladder.py > python-paillier_amd/phe/ladder_gfx950.txt        (on the GPU box, ~2 minutes)
    python tools/calibrate_ladder.py --check python-paillier_amd/phe/ladder_gfx950.txt  (picks of the table vs the pinned best)

Families: 1 = fixed exponent (raw_encrypt's r^n, phe/paillier.py:137), 2 = the CRT halves of raw_decrypt (:347, :351), 3 = per-element
exponents (_raw_mul, :751; 56-bit scalars: the encoding of a float).  A point is `reps` launches queued back to back on one stream
between two HIP events.  A rung that is 2.5x slower than the best one at some size, and wider than it, is not measured at larger
sizes (the library scales its last point with the rows: it never wins there).  Lines: "key_bits family G rows ns".
MEASUREMENT TOOL: never imported by the product; the product only reads the text it wrote."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "python-paillier_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)

FAMILIES = {1: "encrypt", 2: "decrypt", 3: "mul"}


def key(bits):
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "paillier_%d.json" % bits)))
    return {k: int(g[k], 16) for k in ("n", "p", "q", "hp", "hq", "p_inverse")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--key-bits", type=int, nargs="+", default=[1024, 2048, 3072])
    ap.add_argument("--min", type=int, default=7)
    ap.add_argument("--max", type=int, default=17)
    ap.add_argument("--budget-ms", type=float, default=120.0, help="device time per point (sets the repetitions, at least 2)")
    ap.add_argument("--check", default=None, help="a table: report, per size, the rung it picks and the best pinned rung's time")
    args = ap.parse_args()
    import torch
    from phe import _native as native
    from csrc_hash import csrc_hash
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream()
    st = stream.cuda_stream
    out_lines = []
    say = lambda s: (out_lines.append(s), sys.stdout.write(s + "\n"), sys.stdout.flush())
    if not args.check:
        say("# measured geometry ladder: key_bits family G rows ns   (family 1 encrypt r^n, 2 CRT halves of decrypt, 3 _raw_mul 56-bit)")
        prop = torch.cuda.get_device_properties(0)
        say("# device: %s   arch: %s   cus: %d   made: %s   csrc_sha256: %s"
            % (torch.cuda.get_device_name(0), getattr(prop, "gcnArchName", "gfx950").split(":")[0], prop.multi_processor_count,
               time.strftime("%Y-%m-%d"), csrc_hash()))
    report = {}
    for bits in args.key_bits:
        k = key(bits)
        s1, s2 = bits // 32, bits // 16
        ctx = native.Context(k["n"], k["p"], k["q"], k["hp"], k["hq"], k["p_inverse"], n_limbs=s1)
        ctx.load_ladder(None)
        top = 1 << args.max
        gen = torch.Generator(device=dev)
        gen.manual_seed(7)
        rnd = lambda rows, cols: torch.randint(-2 ** 31, 2 ** 31, (rows, cols), dtype=torch.int32, device=dev, generator=gen)
        m, r = rnd(top, s1), rnd(top, s1)
        m[:, s1 - 1] = 0
        r[:, s1 - 1] &= 0x3fffffff
        r[:, 0] |= 1
        e = rnd(top, 2)
        e[:, 1] &= 0x00ffffff
        c = torch.empty((top, s2), dtype=torch.int32, device=dev)
        out = torch.empty_like(c)
        back = torch.empty((top, s1), dtype=torch.int32, device=dev)
        ctx.encrypt_dev(m.data_ptr(), r.data_ptr(), c.data_ptr(), top, st)
        torch.cuda.synchronize()
        fns = {
            "encrypt": lambda B: ctx.encrypt_dev(m.data_ptr(), r.data_ptr(), out.data_ptr(), B, st),
            "decrypt": lambda B: ctx.decrypt_dev(c.data_ptr(), back.data_ptr(), B, st),
            "mul": lambda B: ctx.powmod_dev(c.data_ptr(), e.data_ptr(), 2, 56, out.data_ptr(), B, st),
        }

        def timed(fn, B):
            fn(B)
            stream.synchronize()
            t0 = time.perf_counter()
            fn(B)
            stream.synchronize()
            one_ms = (time.perf_counter() - t0) * 1e3
            reps = int(max(2, min(200, args.budget_ms / max(one_ms, 1e-3))))
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(reps):
                fn(B)
            b.record(stream)
            b.synchronize()
            return a.elapsed_time(b) / reps

        pub, priv = ctx.ladder()
        for fam, op in FAMILIES.items():
            groups = sorted({g // 100 for g in (priv if fam == 2 else pub)})
            geom_key = "geom_priv" if fam == 2 else "geom_pub"
            if args.check:
                ctx.load_ladder(open(args.check).read())
                rows = []
                for lg in range(args.min, args.max + 1):
                    B = 1 << lg
                    ctx.set_group(0)
                    ms_pick = timed(fns[op], B)
                    picked = ctx.last_launch()[geom_key]
                    best = (None, 1e30)
                    for G in groups:
                        ctx.set_group(G)
                        ms = timed(fns[op], B)
                        if ms < best[1]:
                            best = (ctx.last_launch()[geom_key], ms)
                    ctx.set_group(0)
                    rows.append({"log2_batch": lg, "picked": picked, "picked_ms": ms_pick, "best_pinned": best[0], "best_pinned_ms": best[1],
                                 "picked_over_best": ms_pick / best[1]})
                report["%d/%s" % (bits, op)] = rows
                continue
            active = {G: True for G in groups}
            for lg in range(args.min, args.max + 1):
                B = 1 << lg
                got = {}
                for G in groups:
                    if not active[G]:
                        continue
                    ctx.set_group(G)
                    ms = timed(fns[op], B)
                    ran = ctx.last_launch()[geom_key] // 100
                    if ran != G:            # (no rung of exactly this width in this family: the next wider one answered, measured under its own G)
                        continue
                    got[G] = ms
                    say("%d %d %d %d %.0f" % (bits, fam, G, B, ms * 1e6))
                if got:
                    best_g = min(got, key=got.get)
                    for G, ms in got.items():
                        if G > best_g and ms > 2.5 * got[best_g]:
                            active[G] = False
            ctx.set_group(0)
        ctx.close()
        del m, r, e, c, out, back
        torch.cuda.empty_cache()
    if args.check:
        worst = max(row["picked_over_best"] for rows in report.values() for row in rows)
        print(json.dumps({"table": args.check, "worst_picked_over_best_pinned": worst, "ops": report}))


if __name__ == "__main__":
    main()
