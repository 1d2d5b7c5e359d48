#!/usr/bin/env python3
"""The reference's benchmark (examples/benchmarks.py:38-92: encrypt, decrypt, add a float, add a ciphertext, add 1.0,
multiply by a float — per key size from 128 to 8192 bits) on this package, in the two forms a user can write:

  scalar   the reference's own idiom, one Python object per number: `[pub.encrypt(x) for x in xs]`,
           `[priv.decrypt(e) for e in encs]`, `[a + b for a, b in ...]` — every call a batch of one on the GPU
           (encryptions draw their obfuscators r^n from a pool that is refilled by the launch);
  batched  the same work through `encrypt_batch` / `EncryptedVector` operators / `decrypt_batch` on resident vectors.

Both keys of the pair live in the process (as in the reference's script), so once the private key has been used the
encryptions take the key owner's CRT form (same ciphertexts, about half the work); the first scalar
column entry ("encrypt") is measured before that, on the public path with pooled obfuscators.

Prints the reference's table per key size and one JSON object at the end.  Needs an MI355X.

    python examples/benchmarks_batched.py [--scalar-ops 300] [--batch 16384] [--key-sizes 128 256 ... 8192]
"""
import argparse
import gc
import json
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-paillier_amd"))
import phe as paillier  # noqa: E402

OPS = ["encrypt", "decrypt", "add unencrypted and encrypted", "add encrypted and encrypted", "add encrypted and 1",
       "multiply encrypted and unencrypted"]


def timed(fn, best_of=1):
    """wall time of fn() (the shorter of `best_of` calls) and its result"""
    best = None
    for _ in range(best_of):
        t0 = time.perf_counter()
        out = fn()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return best, out


def bench_key(key_size, scalar_ops, batch, rng):
    t_key, (pub, priv) = timed(lambda: paillier.generate_paillier_keypair(n_length=key_size))
    xs = [rng.random() for _ in range(scalar_ops)]
    ys = [rng.random() for _ in range(scalar_ops)]
    pub.encrypt(0.5)                                           # context creation and first launches stay outside
    res = {"key_bits": key_size, "keygen_s": t_key, "scalar": {}, "batched": {}}
    # ---- the reference's idiom --------------------------------------------------------------------------------
    t, enc1 = timed(lambda: [pub.encrypt(x) for x in xs])
    res["scalar"][OPS[0]] = t / scalar_ops
    enc2 = [pub.encrypt(y) for y in ys]
    t, dec = timed(lambda: [priv.decrypt(e) for e in enc1])
    assert dec == xs
    res["scalar"][OPS[1]] = t / scalar_ops
    t, s1 = timed(lambda: [e + y for e, y in zip(enc1, ys)])
    res["scalar"][OPS[2]] = t / scalar_ops
    t, s2 = timed(lambda: [a + b for a, b in zip(enc1, enc2)])
    res["scalar"][OPS[3]] = t / scalar_ops
    t, s3 = timed(lambda: [e + 1.0 for e in enc1])
    res["scalar"][OPS[4]] = t / scalar_ops
    t, s4 = timed(lambda: [e * y for e, y in zip(enc1, ys)])
    res["scalar"][OPS[5]] = t / scalar_ops
    k = min(16, scalar_ops)
    assert all(abs(priv.decrypt(a) - (x + y)) < 1e-9 for a, x, y in zip(s2[:k], xs, ys))
    assert all(abs(priv.decrypt(a) - x * y) < 1e-9 for a, x, y in zip(s4[:k], xs, ys))
    # ---- the batched API on resident vectors -------------------------------------------------------------------
    X, Y = np.array([rng.random() for _ in range(batch)]), np.array([rng.random() for _ in range(batch)])
    # (the previous key size's key pair, engine and GPU context are cyclic garbage by now: collect them here, not when the
    # collector happens to run inside a timed call — freeing a context's scratch buffers takes tens of milliseconds)
    gc.collect()
    # one untimed pass at the batch size that is timed: the first launch of each kernel (code object load) and the first growth of
    # the context's scratch buffers stay outside — a batch of 64 takes other kernels (the small-batch rungs) than one of 16384
    w1, w2 = pub.encrypt_batch(X, device=True), pub.encrypt_batch(Y, device=True)
    for warm in (lambda: priv.decrypt_batch(w1), lambda: (w1 + Y).limbs(False), lambda: (w1 + w2).limbs(False),
                 lambda: (w1 * Y).limbs(False)):
        warm()
    del w1, w2
    pub.discard_obfuscators()                                  # time real encryptions: r drawn and r^n computed in the call
    # (batched column: the shorter of two calls each — a 1 ms call is at the mercy of whatever else the process does once, e.g.
    #  the teardown of the previous key size's context: profiles/r03y_benchmarks_batched.txt had one 28 ms addition at 1024 bits)
    t, v1 = timed(lambda: pub.encrypt_batch(X, device=True), best_of=2)
    res["batched"][OPS[0]] = t / batch
    v2 = pub.encrypt_batch(Y, device=True)
    t, back = timed(lambda: priv.decrypt_batch(v1), best_of=2)
    assert back == X.tolist()
    res["batched"][OPS[1]] = t / batch
    t, a1 = timed(lambda: (v1 + Y).limbs(False), best_of=2)
    res["batched"][OPS[2]] = t / batch
    t, a2 = timed(lambda: (v1 + v2).limbs(False), best_of=2)
    res["batched"][OPS[3]] = t / batch
    t, a3 = timed(lambda: (v1 + 1.0).limbs(False), best_of=2)
    res["batched"][OPS[4]] = t / batch
    t, a4 = timed(lambda: (v1 * Y).limbs(False), best_of=2)
    res["batched"][OPS[5]] = t / batch
    got = priv.decrypt_batch((v1 + v2)[:64])
    assert all(abs(g - (x + y)) < 1e-9 for g, x, y in zip(got, X[:64], Y[:64]))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scalar-ops", type=int, default=300)
    ap.add_argument("--batch", type=int, default=1 << 14)
    ap.add_argument("--key-sizes", type=int, nargs="+", default=[128, 256, 512, 1024, 2048, 4096, 8192])
    args = ap.parse_args()
    rng = random.Random(1)
    out = []
    for ks in args.key_sizes:
        batch = args.batch if ks <= 4096 else max(1024, args.batch // 8)
        r = bench_key(ks, args.scalar_ops, batch, rng)
        out.append(r)
        print("Paillier Benchmarks with key size of %d bits (key pair in %.2f s)" % (ks, r["keygen_s"]))
        print("%-38s %26s %30s" % ("operation", "scalar idiom: s (ops/s)", "batched x%d: s (ops/s)" % batch))
        for op in OPS:
            a, b = r["scalar"][op], r["batched"][op]
            print("%-38s %14.6f (%9d) %16.9f (%11d)" % (op, a, int(1 / a), b, int(1 / b)))
        sys.stdout.flush()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
