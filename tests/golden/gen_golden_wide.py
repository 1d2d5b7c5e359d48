#!/usr/bin/env python3
"""tests/golden/paillier_8192.json.gz: a small set of vectors under an 8192-bit key — the widest key the reference's own
benchmark times (examples/benchmarks.py:88-90) — produced by the REAL reference (imported from /root/reference, CPython-int
engine), like gen_golden.py but trimmed: at this width one pure-Python raw_encrypt takes seconds.

    python tests/golden/gen_golden_wide.py          (build container only; ~5 minutes)

Primes: seeded candidates tested with the reference's own is_prime (phe/util.py:381-443)."""
import gzip
import json
import os
import random
import sys

sys.path.insert(0, "/root/reference")
import phe  # noqa: E402
from phe import paillier, util  # noqa: E402

assert phe.__file__.startswith("/root/reference") and not util.HAVE_GMP
HERE = os.path.dirname(os.path.abspath(__file__))
h = lambda x: format(x, "x")


def seeded_prime(rng, bits):
    while True:
        cand = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
        if util.is_prime(cand):
            return cand


def main(key_bits=8192, seed=15):
    rng = random.Random(seed)
    while True:
        p, q = seeded_prime(rng, key_bits // 2), seeded_prime(rng, key_bits // 2)
        if p != q and (p * q).bit_length() == key_bits:
            break
    n = p * q
    pub = paillier.PaillierPublicKey(n)
    priv = paillier.PaillierPrivateKey(pub, p, q)
    nsq, max_int = pub.nsquare, pub.max_int
    out = {"reference": "data61/python-paillier %s (CPython-int engine)" % phe.__version__, "key_bits": key_bits, "seed": seed,
           "p": h(priv.p), "q": h(priv.q), "n": h(n), "hp": h(priv.hp), "hq": h(priv.hq), "p_inverse": h(priv.p_inverse),
           "max_int": h(max_int)}
    ms = [0, max_int, n - 1, n + 1, rng.randrange(0, n), rng.getrandbits(64)]
    rs = [1, n - 1] + [rng.randrange(1, n) for _ in range(len(ms) - 2)]
    enc = [{"m": h(m), "r": h(r), "c": h(pub.raw_encrypt(m, r_value=r))} for m, r in zip(ms, rs)]
    out["raw_encrypt"] = enc
    cts = [int(e["c"], 16) for e in enc]
    dec = [{"c": e["c"], "m": h(priv.raw_decrypt(int(e["c"], 16)))} for e in enc]
    for _ in range(2):
        c = rng.randrange(1, nsq)
        dec.append({"c": h(c), "m": h(priv.raw_decrypt(c))})
    out["raw_decrypt"] = dec
    obf = []
    for c in cts[:2]:
        r = rng.randrange(1, n)
        obf.append({"c_in": h(c), "r": h(r), "c_out": h(util.mulmod(c, util.powmod(r, n, nsq), nsq))})
    out["obfuscate"] = obf
    holder = paillier.EncryptedNumber(pub, 0, 0)
    out["raw_add"] = [{"a": h(a), "b": h(b), "out": h(holder._raw_add(a, b))}
                      for a, b in [(cts[0], cts[1]), (nsq - 1, nsq - 1), (cts[4], cts[5])]]
    scalars = [0, 1, 16 ** 5, rng.getrandbits(56), n - max_int - 1, n - max_int, n - 1, n - rng.getrandbits(56) - 1]
    out["raw_mul"] = [{"c": h(cts[(i * 5 + 3) % len(cts)]), "s": h(s),
                       "out": h(paillier.EncryptedNumber(pub, cts[(i * 5 + 3) % len(cts)], 0)._raw_mul(s))}
                      for i, s in enumerate(scalars)]
    api = []
    for v in [42, -(2 ** 40), 3.141592653, -4.6e-12]:
        r = rng.randrange(1, n)
        en = pub.encrypt(v, r_value=r)
        api.append({"value": repr(v), "r": h(r), "c": h(en.ciphertext(False)), "exponent": en.exponent,
                    "decrypted": repr(priv.decrypt(en))})
    out["encrypt_api"] = api
    path = os.path.join(HERE, "paillier_%d.json.gz" % key_bits)
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(json.dumps(out, indent=0).encode())
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
