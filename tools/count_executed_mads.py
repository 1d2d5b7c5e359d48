#!/usr/bin/env python3
"""Exact multiply-add counts of the kernels, from the code itself.

Every `wave::mad64` call of the device headers is one `v_mad_u64_u32` lane-operation on gfx950.  The CPU wave emulator
(tests/emu: the SAME headers compiled for the host) counts those calls, so running one full wavefront's worth of
elements through it gives the multiply-adds a kernel EXECUTES per element — lane-operations, the unit of the VALU
roofline — without a hand model.  bench.py reads the committed result (profiles/executed_mads_r*.json) for
`roofline.executed` / `roofline.frac`; profiles/README.md says how it relates to the PMC count SQ_INSTS_VALU.

    python tools/count_executed_mads.py --out profiles/executed_mads_r02.json [--key-bits 1024 2048 3072]

TEST/MEASUREMENT INFRASTRUCTURE: runs on the CPU box (no GPU), never imported by the product path."""
import argparse
import ctypes
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "python-paillier_amd")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--key-bits", type=int, nargs="+", default=[1024, 2048, 3072])
    ap.add_argument("--elements", type=int, default=64, help="elements per run: a multiple of 64/G for every geometry")
    args = ap.parse_args()
    from emu_lib import Emu
    from phe import _native as native
    emu = Emu()
    emu.L.emu_mad_count.restype = ctypes.c_uint64
    count = lambda: int(emu.L.emu_mad_count(1))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from csrc_hash import csrc_hash
    out = {"unit": "v_mad_u64_u32 lane-operations per element (wave::mad64 calls counted by the CPU wave emulator on "
                   "full wavefronts; 29-bit limbs)", "elements_per_run": args.elements,
           "csrc_sha256": csrc_hash(), "keys": {}}
    for bits in args.key_bits:
        g = json.load(open(os.path.join(ROOT, "tests", "golden", "paillier_%d.json" % bits)))
        H = lambda k: int(g[k], 16)
        n, p, q = H("n"), H("p"), H("q")
        s1, s2, sh = bits // 32, bits // 16, bits // 64
        rng = random.Random(bits)
        B = args.elements
        n_arr = native.int_to_limbs(n, s1)
        nsq_arr = native.int_to_limbs(n * n, s2)
        m = native.ints_to_limbs([rng.randrange(n) for _ in range(B)], s1)
        r = native.ints_to_limbs([rng.randrange(1, n) for _ in range(B)], s1)
        res = {"split_geometry_GL": list(emu.split_geometry(n_arr))}
        t0 = time.time()
        count()
        c = emu.encrypt(n_arr, m, r)
        res["encrypt"] = count() / B
        assert native.limbs_to_ints(c[:1])[0] == (1 + n * native.limbs_to_ints(m[:1])[0]) * pow(
            native.limbs_to_ints(r[:1])[0], n, n * n) % (n * n)
        emu.obfuscate(n_arr, c, r)
        res["obfuscate"] = count() / B
        key_o = [native.int_to_limbs(v, sh) for v in (p, q, H("hp"), H("hq"), H("p_inverse"))]
        co = emu.encrypt_owner(n_arr, *key_o, m, r)
        if co is not None:
            assert np.array_equal(co, c)
            res["encrypt_key_owner"] = count() / B
        else:
            count()
        key = [native.int_to_limbs(v, sh) for v in (p, q, H("hp"), H("hq"), H("p_inverse"))]
        # the CRT tail as the device runs it: one wavefront per ciphertext from 1024-bit keys up (phe_hip.hip, create_private),
        # one ciphertext per thread (plain C arithmetic, no wave::mad64: not counted) below
        wave_tail = bits >= 1024
        emu.set_wave_tail(wave_tail)
        try:
            back = emu.decrypt(*key, s1, c)
        finally:
            emu.set_wave_tail(False)
        res["decrypt"] = count() / B
        res["decrypt_tail"] = "wavefront" if wave_tail else "thread"
        assert np.array_equal(back, m)
        # the same per geometry of the CRT halves: the ladder may end on another rung than the narrowest (3072 bits: 4 x 14,
        # not 2 x 27, since the wide-rung factor of rung_cost) — bench.py looks the count up by what last_launch reported
        by_geom = {}
        for group in (0, 1, 4):
            emu.set_group(group)
            emu.set_wave_tail(wave_tail)
            try:
                G, L = emu.split_geometry(native.int_to_limbs(q, sh))
                back = emu.decrypt(*key, s1, c)
            finally:
                emu.set_wave_tail(False)
                emu.set_group(0)
            by_geom[str(G * 100 + L)] = count() / B
            assert np.array_equal(back, m)
        res["decrypt_by_halves_geometry"] = by_geom
        # _raw_add as the library runs it on large batches: two Montgomery products, or — where the key's table is offered —
        # one plain product + one fold (mul_table.h; by tiles with one element per lane: mul_tile.h); every form is counted
        c_rev = np.ascontiguousarray(c[::-1])
        want = emu.mulmod(nsq_arr, c, c_rev)
        res["raw_add_two_montgomery_products"] = count() / B
        offered = int(emu.L.emu_table_mul_offered(nsq_arr.ctypes.data_as(ctypes.c_void_p), s2))
        res["raw_add_form"] = "tiles" if offered & 2 else ("table_in_lds" if offered & 1 else "two_montgomery_products")
        if offered & 1:
            assert np.array_equal(emu.mulmod_table(nsq_arr, c, c_rev), want)
            res["raw_add_table_in_lds"] = count() / B
        if offered & 2:
            waves = 8 if offered & 4 else 16          # (the 512-thread workgroup shape: n^2 of a 1024-bit key in 8 x 9 columns)
            assert np.array_equal(emu.mulmod_table(nsq_arr, c, c_rev, tiles=True, blocks=1, waves=waves), want)
            res["raw_add_tiles"] = count() / B
            res["raw_add_tile_waves"] = waves
        res["raw_add"] = res["raw_add_" + res["raw_add_form"]]
        emu.add_plain(n_arr, c, m)
        res["add_plain"] = count() / B
        for name, ebits in (("raw_mul_56bit", 56), ("raw_mul_63bit", 63)):
            e = native.ints_to_limbs([rng.getrandbits(ebits) | (1 << (ebits - 1)) for _ in range(B)], 2)
            emu.powmod_n2(n_arr, c, e)
            res[name] = count() / B
        res["seconds"] = round(time.time() - t0, 1)
        out["keys"][str(bits)] = res
        print(bits, res, flush=True)
    text = json.dumps(out, indent=1)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text + "\n")
    else:
        print(text)


if __name__ == "__main__":
    main()
