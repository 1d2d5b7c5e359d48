#!/usr/bin/env python3
"""Record every hot-path call the REFERENCE'S OWN TEST SUITES make, as a replayable trace.

Run in the build container only (needs /root/reference):

    python tests/golden/gen_reference_trace.py [default_key_bits=2048]

The real reference (data61/python-paillier 1.5.0, imported unmodified from /root/reference) runs its own unittest
suites phe/tests/paillier_test.py and phe/tests/math_test.py.  The five functions of the hot path are wrapped — nothing
else is touched — and every successful call is written down with its operands and the value the reference returned:

    PaillierPublicKey.raw_encrypt(m, r)   phe/paillier.py:102-139   ("enc": m, r, c;  r = r_value or the value drawn)
    EncryptedNumber.obfuscate()           :603-624                   ("obf": c_in, r drawn, c_out)
    PaillierPrivateKey.raw_decrypt(c)     :328-354                   ("dec": c, m)
    EncryptedNumber._raw_add(a, b)        :705-719                   ("add": a, b, out)
    EncryptedNumber._raw_mul(k)           :721-751                   ("mul": c, k, out — both branches occur)

The GPU box has no /root/reference; tests/test_reference_trace.py replays the trace through libphe_hip.so there and
demands the same bits (and, without a GPU, through the CPU restatements in oracle/, which pins the oracle to the suite).
The default key size is lowered from 3072 to 2048 bits (the suites call generate_paillier_keypair() without arguments);
everything a test asks for explicitly is left as the test wrote it.  Numbers are lower-case hex.
"""
import gzip
import json
import os
import sys
import unittest

sys.path.insert(0, "/root/reference")
import phe  # noqa: E402
from phe import paillier, util  # noqa: E402

assert phe.__file__.startswith("/root/reference"), phe.__file__
assert not util.HAVE_GMP

HERE = os.path.dirname(os.path.abspath(__file__))
BITS = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
paillier.generate_paillier_keypair.__defaults__ = (None, BITS)

h = lambda x: format(x, "x")
state = {"test": None}
tests, keys, key_index, ops, seen = [], [], {}, [], set()


def key_of(pub, priv=None):
    n = pub.n
    if n not in key_index:
        key_index[n] = len(keys)
        keys.append({"n": h(n), "p": None, "q": None})
    k = key_index[n]
    if priv is not None and keys[k]["p"] is None:
        keys[k]["p"], keys[k]["q"] = h(priv.p), h(priv.q)
    return k


def record(op, k, *vals):
    row = (op, k) + tuple(h(v) for v in vals)
    t = state["test"]
    if t is None:
        return
    if (t, row) in seen:                             # identical call again in the same test: once is enough
        return
    seen.add((t, row))
    ops.append([t] + list(row))


orig_enc = paillier.PaillierPublicKey.raw_encrypt
orig_dec = paillier.PaillierPrivateKey.raw_decrypt
orig_add = paillier.EncryptedNumber._raw_add
orig_mul = paillier.EncryptedNumber._raw_mul
orig_obf = paillier.EncryptedNumber.obfuscate


def raw_encrypt(self, plaintext, r_value=None):
    if not isinstance(plaintext, int):
        return orig_enc(self, plaintext, r_value)    # raises TypeError as the reference does
    r = r_value or self.get_random_lt_n()            # phe/paillier.py:136, drawn here so that it can be written down
    c = orig_enc(self, plaintext, r)
    record("enc", key_of(self), plaintext, r, c)
    return c


def raw_decrypt(self, ciphertext):
    m = orig_dec(self, ciphertext)
    record("dec", key_of(self.public_key, self), ciphertext, m)
    return m


def raw_add(self, e_a, e_b):
    out = orig_add(self, e_a, e_b)
    record("add", key_of(self.public_key), e_a, e_b, out)
    return out


def raw_mul(self, plaintext):
    out = orig_mul(self, plaintext)
    record("mul", key_of(self.public_key), self.ciphertext(False), plaintext, out)
    return out


def obfuscate(self):
    pk = self.public_key
    c0 = self.ciphertext(False)
    r = pk.get_random_lt_n()
    pk.get_random_lt_n = lambda: r                   # instance attribute shadows the method for this one call
    try:
        orig_obf(self)
    finally:
        del pk.get_random_lt_n
    if isinstance(c0, int):
        record("obf", key_of(pk), c0, r, self.ciphertext(False))


paillier.PaillierPublicKey.raw_encrypt = raw_encrypt
paillier.PaillierPrivateKey.raw_decrypt = raw_decrypt
paillier.EncryptedNumber._raw_add = raw_add
paillier.EncryptedNumber._raw_mul = raw_mul
paillier.EncryptedNumber.obfuscate = obfuscate


class Result(unittest.TextTestResult):
    def startTest(self, test):
        super().startTest(test)
        name = test.id()
        if name not in tests:
            tests.append(name)
        state["test"] = tests.index(name)

    def stopTest(self, test):
        super().stopTest(test)
        state["test"] = None


def main():
    from phe.tests import math_test, paillier_test
    suite = unittest.TestSuite()
    for mod in (paillier_test, math_test):
        suite.addTests(unittest.defaultTestLoader.loadTestsFromModule(mod))
    runner = unittest.TextTestRunner(resultclass=Result, verbosity=1)
    res = runner.run(suite)
    assert res.wasSuccessful(), "the reference's own suite must pass on the reference"
    used = sorted({row[0] for row in ops})
    doc = {"reference": "data61/python-paillier 1.5.0, phe/tests/paillier_test.py + math_test.py run on the reference itself",
           "default_key_bits": BITS, "tests_run": res.testsRun, "tests": tests, "tests_with_hot_path_calls": len(used),
           "keys": keys, "ops": ops}
    path = os.path.join(HERE, "reference_suite_trace.json.gz")
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(json.dumps(doc, separators=(",", ":")).encode())
    counts = {}
    for row in ops:
        counts[row[1]] = counts.get(row[1], 0) + 1
    print("tests run %d, with hot-path calls %d, keys %d, ops %s -> %s (%d bytes)"
          % (res.testsRun, len(used), len(keys), counts, path, os.path.getsize(path)))


if __name__ == "__main__":
    main()
