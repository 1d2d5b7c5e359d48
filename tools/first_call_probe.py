#!/usr/bin/env python3
"""Is the first LARGE call of a process slower than the later ones, and does a run of scalar calls before it matter?
(examples/benchmarks_batched.py times ONE call per operation; its batched column moved between runs.)
    python tools/first_call_probe.py [scalar_calls_before]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-paillier_amd"))
import numpy as np
import phe as paillier
pub, priv = paillier.generate_paillier_keypair(n_length=2048)
X = np.random.default_rng(1).random(16384)
before = int(sys.argv[1]) if len(sys.argv) > 1 else 0
t0 = time.perf_counter()
enc = [pub.encrypt(float(x)) for x in X[:before]]
dec = [priv.decrypt(e) for e in enc]
sums = [a + b for a, b in zip(enc, enc[1:])]
prods = [a * 3.5 for a in enc]
print("%d scalar encrypt/decrypt/add/mul calls first: %.2f s" % (before, time.perf_counter() - t0))
w1 = pub.encrypt_batch(X[:64], device=True)
priv.decrypt_batch(w1)
pub.discard_obfuscators()
eng = priv._get_engine()
for i in range(3):
    t0 = time.perf_counter()
    v = pub.encrypt_batch(X, device=True)
    t1 = time.perf_counter()
    back = priv.decrypt_batch(v)
    t2 = time.perf_counter()
    print("large call %d: encrypt_batch %.1f ms (%.0f/s)  decrypt_batch %.1f ms (%.0f/s)  %s" % (
        i, (t1 - t0) * 1e3, 16384 / (t1 - t0), (t2 - t1) * 1e3, 16384 / (t2 - t1), eng.ctx.last_launch()))
    pub.discard_obfuscators()
