#!/usr/bin/env python3
"""the REFERENCE ITSELF as it actually runs — one process, one Python object per number — on CPU,
next to the all-core libgmp port that bench.py reports as `cpu_baseline`.

Runs only where /root/reference exists (the build container; the GPU box has no reference and this file is never imported by
bench.py, the tests or the product).  The reference is imported as it lies, read-only; its three backend functions
phe.util.powmod / mulmod / invert (phe/util.py:38-50, :53-64, :85-103) are bound to the libgmp calls gmpy2 would make
(`mpz_powm`, `mpz_mul` + `mpz_mod`, `mpz_invert` through oracle/libphe_oracle.so) because gmpy2 itself is not installable
here — the names are patched in phe.paillier too, which binds them at import (phe/paillier.py:29).  A second column runs
the same loop on the pure-Python fallback the reference takes without gmpy2 (phe/util.py:48, :61, :100-103).

    python tools/ref_cpu_baseline.py [--key-bits 2048] [--ops 200]  > profiles/rNN_reference_single_process_cpu.json
"""
import argparse
import json
import os
import platform
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--key-bits", type=int, default=2048)
    ap.add_argument("--ops", type=int, default=200)
    args = ap.parse_args()
    if not os.path.isdir(os.path.join(REF, "phe")):
        sys.exit("needs the reference checkout at %s (the build container)" % REF)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, REF)
    import phe
    import phe.paillier
    import phe.util
    assert phe.__file__.startswith(REF), phe.__file__
    from oracle.paillier_oracle import COracle
    orc = COracle()
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "paillier_%d.json" % args.key_bits)))
    H = lambda k: int(g[k], 16)
    pub = phe.paillier.PaillierPublicKey(H("n"))
    priv = phe.paillier.PaillierPrivateKey(pub, H("p"), H("q"))
    rng = random.Random(3)
    xs = [rng.random() for _ in range(args.ops)]

    def run():
        t0 = time.perf_counter()
        enc = [pub.encrypt(x) for x in xs]                       # examples/benchmarks.py:47-49
        t1 = time.perf_counter()
        dec = [priv.decrypt(e) for e in enc]                     # :51-53
        t2 = time.perf_counter()
        sums = [a + b for a, b in zip(enc, enc[1:] + enc[:1])]   # :62-65
        t3 = time.perf_counter()
        prods = [e * x for e, x in zip(enc, xs)]                 # :69-71
        t4 = time.perf_counter()
        assert dec == xs
        for s in sums[:4] + prods[:4]:
            s.ciphertext(be_secure=False)
        return {"encrypts_per_s": args.ops / (t1 - t0), "decrypts_per_s": args.ops / (t2 - t1),
                "adds_per_s": args.ops / (t3 - t2), "scalar_muls_per_s": args.ops / (t4 - t3)}

    fallback = run()                                             # HAVE_GMP is False here: CPython pow, the reference's own fallback
    originals = {m: {f: getattr(m, f) for f in ("powmod", "mulmod", "invert")} for m in (phe.util, phe.paillier)}

    def powmod(a, b, c):
        if a == 1:                                               # phe/util.py:45-46
            return 1
        if max(a, b, c) < (1 << 16):                             # :47-48
            return pow(a, b, c)
        return orc.powmod(a, b, c)

    def mulmod(a, b, c):
        if max(a, b, c) < (1 << 1000):                           # :60-61
            return a * b % c
        return orc.mulmod(a, b, c)

    def invert(a, b):
        return orc.invert(a, b)
    for mod in originals:
        mod.powmod, mod.mulmod, mod.invert = powmod, mulmod, invert
    try:
        gmp = run()
    finally:
        for mod, fns in originals.items():
            for name, fn in fns.items():
                setattr(mod, name, fn)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count()
    print(json.dumps({
        "what": "the reference (data61/python-paillier, imported read-only from %s) in ONE process, the loop of its "
                "examples/benchmarks.py, %d-bit key, %d operations per kind" % (REF, args.key_bits, args.ops),
        "host": "BUILD CONTAINER cpu (not the GPU box): %s, %d cores visible" % (platform.processor() or platform.machine(), cores),
        "reference_with_libgmp_backend": dict(gmp, note="powmod/mulmod/invert bound to libgmp %s through a ctypes shim — the calls "
                                              "gmpy2 makes; includes ~10 us of int<->limb conversion per call that gmpy2 does not pay" % orc.gmp_version),
        "reference_pure_python_fallback": dict(fallback, note="HAVE_GMP False: CPython pow(), what the reference runs without gmpy2"),
        "key_bits": args.key_bits}))


if __name__ == "__main__":
    main()
