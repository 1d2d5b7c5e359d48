#!/usr/bin/env python3
"""Latency of the scalar API and of tiny batches (the shape of the reference's own benchmark, examples/benchmarks.py:12-29: one
operation at a time): priv.decrypt / pub.raw_encrypt of ONE number and raw_decrypt_batch of 1 ... 256 numbers, per key size, with
the wave-pair kernels and without them (PHE_HIP_NO_WAVE_PAIRS=1 in a second process).  Wall time per call, host side included.

    python tools/bench_latency.py [--key-sizes 1024 2048 3072 4096 8192]  > profiles/rNN_latency.json"""
import argparse
import json
import os
import random
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-paillier_amd"))


def measure(key_sizes):
    import phe as paillier
    rng = random.Random(5)
    out = {}
    for ks in key_sizes:
        pub, priv = paillier.generate_paillier_keypair(n_length=ks)
        eng = priv._get_engine()
        xs = [rng.randrange(pub.n) for _ in range(256)]
        rs = [rng.randrange(1, pub.n) for _ in range(256)]
        cts = pub.raw_encrypt_batch(xs, rs)
        assert priv.raw_decrypt_batch(cts) == xs
        row = {}
        reps = 30 if ks <= 2048 else 8
        priv.raw_decrypt(cts[0])
        t0 = time.perf_counter()
        for i in range(reps):
            assert priv.raw_decrypt(cts[i]) == xs[i]
        row["raw_decrypt_one_ms"] = (time.perf_counter() - t0) / reps * 1e3
        for b in (4, 16, 64, 256):
            priv.raw_decrypt_batch(cts[:b])
            t0 = time.perf_counter()
            for _ in range(max(2, reps // 4)):
                priv.raw_decrypt_batch(cts[:b])
            row["raw_decrypt_batch_%d_ms" % b] = (time.perf_counter() - t0) / max(2, reps // 4) * 1e3
        row["decrypt_path"] = eng.ctx.last_launch()
        # the key owner's encryption (what pub.raw_encrypt of a key PAIR takes: r^n from its CRT halves), one number per call
        pub.raw_encrypt(xs[0], rs[0])
        t0 = time.perf_counter()
        for i in range(max(2, reps // 3)):
            assert pub.raw_encrypt(xs[i], rs[i]) == cts[i]
        row["raw_encrypt_one_key_owner_ms"] = (time.perf_counter() - t0) / max(2, reps // 3) * 1e3
        row["encrypt_key_owner_path"] = eng.ctx.last_launch()
        os.environ["PHE_HIP_OWNER_ENCRYPT"] = "0"              # the public-key path, one exponentiation per call (no pool)
        fresh = paillier.PaillierPublicKey(pub.n)
        fresh.raw_encrypt(xs[0], rs[0])
        t0 = time.perf_counter()
        for i in range(max(2, reps // 3)):
            assert fresh.raw_encrypt(xs[i], rs[i]) == cts[i]
        row["raw_encrypt_one_public_path_ms"] = (time.perf_counter() - t0) / max(2, reps // 3) * 1e3
        row["encrypt_path"] = fresh._get_engine().ctx.last_launch()
        os.environ.pop("PHE_HIP_OWNER_ENCRYPT", None)
        out[str(ks)] = row
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--key-sizes", type=int, nargs="+", default=[1024, 2048, 3072, 4096, 8192])
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    if args.child:
        print(json.dumps(measure(args.key_sizes)))
        return
    res = {}
    for name, env in (("wave_pairs", {}), ("single_wave_kernels", {"PHE_HIP_NO_WAVE_PAIRS": "1"})):
        e = dict(os.environ)
        e.update(env)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--key-sizes"] + [str(k) for k in args.key_sizes],
                           env=e, capture_output=True, text=True)
        if p.returncode:
            sys.stderr.write(p.stderr[-2000:])
            sys.exit(1)
        res[name] = json.loads(p.stdout.strip().splitlines()[-1])
    print(json.dumps(res))
    for ks in args.key_sizes:
        a, b = res["wave_pairs"][str(ks)], res["single_wave_kernels"][str(ks)]
        sys.stderr.write("%5d bits: decrypt one %.3f ms (single-wave kernels %.3f), x16 %.3f (%.3f), x256 %.3f (%.3f); encrypt one %.3f (%.3f), "
                         "by the key owner %.3f (%.3f)\n" % (
            ks, a["raw_decrypt_one_ms"], b["raw_decrypt_one_ms"], a["raw_decrypt_batch_16_ms"], b["raw_decrypt_batch_16_ms"],
            a["raw_decrypt_batch_256_ms"], b["raw_decrypt_batch_256_ms"], a["raw_encrypt_one_public_path_ms"], b["raw_encrypt_one_public_path_ms"],
            a["raw_encrypt_one_key_owner_ms"], b["raw_encrypt_one_key_owner_ms"]))


if __name__ == "__main__":
    main()
