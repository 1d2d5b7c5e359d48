#!/usr/bin/env python3
"""Second source for bench.py's executed multiply-adds: the v_mad_u64_u32 instructions hipcc emits for the
sweep loops of the headline kernel, times their trip counts, against what the CPU wave emulator counts (profiles/executed_mads_r*.json).

k_modexp_split<4,18,encrypt,unit> (2048-bit keys: H = 72 limbs of the scaled modulus on 4 lanes x 18): a pair SQUARE is one loop of
H / 18 = 4 trips, a pair PRODUCT one loop of 4 trips; a wave holds 16 numbers, so multiply-adds per number = instructions per trip x 4
trips x 64 lanes / 16 numbers.  The closed form bench.py:executed_mads uses is (3 + 10/18) H^2 per square (round 5: the symmetric half of
X0*X0, 10 of 18 limbs per lane and row — csrc/split_core.h sq_row; 4 H^2 before) and 5 H^2 per product; the
emulator's count of the whole encryption must then equal squares x 4 H^2 + products x 5 H^2 + entry/exit (tests/test_bench_contract.py
holds the closed form within 1 % of the emulator).  Usage: python tools/static_mad_tally.py [--out profiles/...txt]"""
import argparse
import json
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources as kr  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    meta = kr.kernels(kr.assembly("kernels_s4b"))
    name = [k for k in meta if "k_modexp_split" in k and "halves" not in k and "Lb1EE" in k and "Li0E" in k]
    name = name[0] if name else None
    lines = []
    H, L, per_wave = 72, 18, 16
    sq_limbs = L // 2 + 1                      # limbs a lane takes per row of a squaring's first word (even L)
    want = {"square": (3 * L + sq_limbs) * H * H // L, "product": 5 * H * H}
    label = {"square": "(3 + %d/%d) H^2" % (sq_limbs, L), "product": "5 H^2"}
    ok = False
    if name:
        loops = meta[name]["loops"]
        inner = sorted({(l["mads"], l["insts"]) for l in loops if l["scratch"] == 0 and l["mads"] >= 1000 and l["insts"] < 2200})
        lines.append("kernel %s: %d VGPRs, %d v_mad_u64_u32 in %d instructions" % (kr.short(name), meta[name]["vgpr"], meta[name]["mads"], meta[name]["instructions"]))
        found = {}
        for mads, insts in inner:
            per_number = mads * (H // L) * 64 // per_wave
            kind = [k for k, v in want.items() if abs(per_number / v - 1) < 0.005]
            lines.append("  loop of %4d instructions, %4d multiply-adds per trip -> x %d trips x 64 lanes / %d numbers = %6d per number%s"
                         % (insts, mads, H // L, per_wave, per_number, (" = %s (%s = %d)" % (kind[0], label[kind[0]], want[kind[0]])) if kind else ""))
            for k in kind:
                found[k] = per_number
        ok = set(found) == set(want)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "executed_mads_r*.json")))
    if files:
        rec = json.load(open(files[-1]))["keys"]["2048"]
        # 2048-bit exponent n on the sliding-window schedule of key_setup.h: the emulator's total against squares and products
        lines.append("emulator (%s): %d multiply-adds per encryption = %.1f H^2" % (os.path.basename(files[-1]), rec["encrypt"], rec["encrypt"] / (H * H)))
    try:
        sys.path.insert(0, ROOT)
        import bench
        enc, _ = bench.executed_mads(2048, {"lane_limbs_pub": 418, "lane_limbs_priv": 218, "engine_pub": "split", "engine_priv": "split"})
        lines.append("closed form bench.py:executed_mads = ((3 + 10/18) x 2049 squares + 5 x 325 products + 12 entry/exit) H^2 = %d = %.1f H^2"
                     % (enc, enc / (H * H)))
    except Exception as e:  # noqa: BLE001
        lines.append("closed form not evaluated: %r" % (e,))
    lines.append("static tally agrees with (3 + 10/18) H^2 per square and 5 H^2 per product within 0.5 %%: %s" % ("yes" if ok else "NO"))
    text = "\n".join(lines)
    print(text)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text + "\n")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
