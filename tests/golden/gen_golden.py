#!/usr/bin/env python3
"""Generate tests/golden/*.json by running the REAL reference (data61/python-paillier 1.5.0).

Run in the build container only (needs /root/reference; the GPU box does not have it):

    python tests/golden/gen_golden.py

The reference is imported unmodified from /root/reference.  gmpy2 is not installable
here, so the reference executes its own CPython-int branches (phe/util.py:48, :61,
:100-103); every value stored is a canonical residue, hence engine-independent.
All randomness comes from random.Random(seed) — the reference itself only offers
SystemRandom, so r is always passed explicitly (raw_encrypt(m, r_value=r)).

Numbers are stored as lower-case hex strings (no 0x).
"""
import json
import os
import random
import sys

sys.path.insert(0, "/root/reference")
import phe  # noqa: E402
from phe import paillier, util  # noqa: E402

assert phe.__file__.startswith("/root/reference"), phe.__file__
assert not util.HAVE_GMP

HERE = os.path.dirname(os.path.abspath(__file__))


def seeded_prime(rng, bits):
    """A `bits`-bit prime from a seeded stream, tested with the reference's own is_prime
    (phe/util.py:381-443)."""
    while True:
        cand = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
        if util.is_prime(cand):
            return cand


def make_key(rng, key_bits):
    while True:
        p = seeded_prime(rng, key_bits // 2)
        q = seeded_prime(rng, key_bits // 2)
        if p != q and (p * q).bit_length() == key_bits:
            return p, q


def h(x):
    return format(x, "x")


def gen_for_key(key_bits, seed, n_random):
    rng = random.Random(seed)
    p, q = make_key(rng, key_bits)
    n = p * q
    pub = paillier.PaillierPublicKey(n)
    priv = paillier.PaillierPrivateKey(pub, p, q)
    nsq = pub.nsquare
    max_int = pub.max_int

    out = {
        "reference": "data61/python-paillier %s (CPython-int engine)" % phe.__version__,
        "key_bits": key_bits, "seed": seed,
        "p": h(priv.p), "q": h(priv.q), "n": h(n),
        "hp": h(priv.hp), "hq": h(priv.hq), "p_inverse": h(priv.p_inverse),
        "max_int": h(max_int),
    }

    # --- raw_encrypt (phe/paillier.py:102-139) --------------------------------
    ms = [0, 1, 2, max_int, max_int + 1, n - max_int - 1, n - max_int, n - 2, n - 1, n, n + 1,
          123456789123456789123456789123456789 % n]
    ms += [rng.randrange(0, n) for _ in range(n_random)]
    ms += [rng.getrandbits(64) for _ in range(2)]
    rs = [1, n - 1, 2] + [rng.randrange(1, n) for _ in range(len(ms) - 3)]
    enc = []
    for m, r in zip(ms, rs):
        c = pub.raw_encrypt(m, r_value=r)
        enc.append({"m": h(m), "r": h(r), "c": h(c)})
    out["raw_encrypt"] = enc

    # --- raw_decrypt (phe/paillier.py:328-374) --------------------------------
    dec = []
    for e in enc:
        c = int(e["c"], 16)
        dec.append({"c": e["c"], "m": h(priv.raw_decrypt(c))})
    for _ in range(4):  # arbitrary residues (not produced by encrypt)
        c = rng.randrange(1, nsq)
        dec.append({"c": h(c), "m": h(priv.raw_decrypt(c))})
    out["raw_decrypt"] = dec

    # --- obfuscate (phe/paillier.py:603-624): c * r^n mod n^2 via the same primitives ---
    obf = []
    for e in enc[:4]:
        c = int(e["c"], 16)
        r = rng.randrange(1, n)
        c2 = util.mulmod(c, util.powmod(r, n, nsq), nsq)
        obf.append({"c_in": e["c"], "r": h(r), "c_out": h(c2)})
    out["obfuscate"] = obf

    # --- _raw_add (phe/paillier.py:705-719) -----------------------------------
    cts = [int(e["c"], 16) for e in enc]
    add = []
    pairs = [(cts[0], cts[1]), (cts[8], cts[8]), (1, cts[3]), (nsq - 1, nsq - 1)]
    pairs += [(rng.choice(cts), rng.choice(cts)) for _ in range(n_random)]
    holder = paillier.EncryptedNumber(pub, 0, 0)
    for a, b in pairs:
        add.append({"a": h(a), "b": h(b), "out": h(holder._raw_add(a, b))})
    out["raw_add"] = add

    # --- _raw_mul (phe/paillier.py:721-751), both branches ---------------------
    scalars = [0, 1, 2, 16, 16 ** 5, max_int, n - max_int - 1,   # positive branch (:751)
               n - max_int, n - 1, n - 2, n - 16 ** 3]            # inverse branch  (:745-749)
    scalars += [rng.getrandbits(56) for _ in range(n_random)]
    scalars += [n - rng.getrandbits(56) - 1 for _ in range(n_random)]
    scalars += [rng.randrange(0, n) for _ in range(2)]
    mul = []
    for i, s in enumerate(scalars):
        c = cts[(i * 5 + 3) % len(cts)]
        en = paillier.EncryptedNumber(pub, c, 0)
        mul.append({"c": h(c), "s": h(s), "out": h(en._raw_mul(s))})
    out["raw_mul"] = mul

    # --- object API with encoding (phe/paillier.py:145-194, phe/encoding.py:110-199) ---
    api = []
    values = [0, 1, -1, 42, -(2 ** 40), 3.141592653, -4.6e-12, 1e30, 0.5, 2 ** 52 + 0.5, -1.0 / 3]
    for v in values:
        r = rng.randrange(1, n)
        en = pub.encrypt(v, r_value=r)
        api.append({"value": repr(v), "r": h(r), "c": h(en.ciphertext(False)), "exponent": en.exponent,
                    "decrypted": repr(priv.decrypt(en))})
    out["encrypt_api"] = api
    return out


def main():
    specs = [(256, 11, 6), (1024, 12, 6), (2048, 13, 6), (3072, 14, 4)]
    for key_bits, seed, n_random in specs:
        data = gen_for_key(key_bits, seed, n_random)
        path = os.path.join(HERE, "paillier_%d.json" % key_bits)
        with open(path, "w") as f:
            json.dump(data, f, indent=0, sort_keys=True)
        print("wrote", path, os.path.getsize(path), "bytes")

    # the reference's own known-answer vectors, restated with their source lines
    kat = {
        "source": "phe/tests/paillier_test.py:128-149, phe/tests/util_test.py:29-44",
        "n": 126869, "p": 293, "q": 433, "m": 10100, "r": 74384, "c": 935906717,
        "encrypt_1_r_1": 126870,
        "powmod": [[5, 3, 3, 2], [2, 10, 1000, 24]],
        "invert": [[3, 4, 3]] + [[a, 101, util.invert(a, 101)] for a in range(1, 101)],
    }
    # verify them against the reference right now
    pub = paillier.PaillierPublicKey(kat["n"])
    priv = paillier.PaillierPrivateKey(pub, kat["p"], kat["q"])
    assert pub.raw_encrypt(kat["m"], kat["r"]) == kat["c"]
    assert priv.raw_decrypt(kat["c"]) == kat["m"]
    assert pub.encrypt(1, r_value=1).ciphertext(False) == kat["encrypt_1_r_1"]
    for a, b, c, want in kat["powmod"]:
        assert util.powmod(a, b, c) == want
    kat["hp"], kat["hq"], kat["p_inverse"] = priv.hp, priv.hq, priv.p_inverse
    with open(os.path.join(HERE, "reference_kat.json"), "w") as f:
        json.dump(kat, f, indent=0, sort_keys=True)
    print("wrote reference_kat.json")


if __name__ == "__main__":
    main()
