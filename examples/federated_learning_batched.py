#!/usr/bin/env python3
"""Federated linear regression with Paillier-encrypted gradient aggregation — batched, on the GPU backend.

Same experiment as the reference's examples/federated_learning_with_encryption.py (BASELINE.json configs[4]):
scikit-learn's diabetes data, 5 hospitals, 50 rounds, eta 1.5, numpy seed 43 — but every hospital encrypts its
whole gradient with ONE `encrypt_batch` launch, the running aggregate is an `EncryptedVector` (one `+` = one
kernel launch), and the server decrypts with ONE `decrypt_batch` launch per round.  Because the homomorphic sum
is exact fixed-point arithmetic, the resulting models — and the printed test errors — are the same as the
reference's scalar loops produce (3775.50 for every hospital at the default settings).

    python examples/federated_learning_batched.py [key_length]

On a multi-GPU node one process can use all devices: PHE_HIP_DEVICES=all (phe/fleet.py) makes every key hold one engine per
GPU and cuts large batches contiguously over them.  This example's batches (11 gradient entries) stay on the first device —
a launch of 11 rows is latency-bound — so its output is the same with or without the variable; what does fan out here is the
offline half: `precompute=True` fills every device's obfuscator pool with its share.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-paillier_amd"))

import phe as paillier  # noqa: E402

SEED = 43
N_HOSPITALS, N_ROUNDS, ETA, TEST_SIZE = 5, 50, 1.5, 50


def load_split(n_parts, seed=SEED):
    """Diabetes features + bias column, shuffled; a 50-row hold-out; the rest cut into equal contiguous parts."""
    from sklearn.datasets import load_diabetes
    rng = np.random.RandomState(seed)          # same stream as np.random.seed(seed) + module-level calls
    data = load_diabetes()
    X = np.c_[data.data, np.ones(len(data.target))]
    y = data.target
    order = rng.permutation(len(y))
    X, y = X[order], y[order]
    held_out = rng.choice(len(y), size=TEST_SIZE, replace=False)
    keep = np.ones(len(y), dtype=bool)
    keep[held_out] = False
    X_train, y_train = X[keep], y[keep]
    per = len(y_train) // n_parts
    parts = [(X_train[i * per:(i + 1) * per], y_train[i * per:(i + 1) * per]) for i in range(n_parts)]
    return parts, X[held_out], y[held_out]


class Hospital:
    def __init__(self, X, y, public_key):
        self.X, self.y, self.public_key = X, y, public_key
        self.w = np.zeros(X.shape[1])

    def gradient(self):
        return (self.X @ self.w - self.y) @ self.X / len(self.X)

    def encrypted_gradient(self):
        return self.public_key.encrypt_batch(self.gradient())      # one launch for the whole vector

    def encrypted_gradient_scalar(self):
        """one `encrypt` call per gradient entry: the scalar API, a launch of ONE row each (how python-paillier is usually driven)"""
        return [self.public_key.encrypt(float(x)) for x in self.gradient()]

    def step(self, g):
        self.w -= ETA * g

    def test_error(self, X_test, y_test):
        return float(np.mean((X_test @ self.w - y_test) ** 2))


def run_scalar(key_length=2048, n_rounds=N_ROUNDS, verbose=True, keypair=None):
    """The same protocol through the drop-in's SCALAR API only — `public_key.encrypt(x)` per gradient entry, `a + b` per pair of
    EncryptedNumbers, `private_key.decrypt(e)` per aggregate entry: 55 + 44 + 11 one-row calls per round, 2,750 + 2,200 + 550 per
    run — the call shape of a program written against python-paillier, nothing batched by the caller.  Returns (errors,
    seconds, calls)."""
    parts, X_test, y_test = load_split(N_HOSPITALS)
    public_key, private_key = keypair or paillier.generate_paillier_keypair(n_length=key_length)
    hospitals = [Hospital(X, y, public_key) for X, y in parts]
    calls = {"encrypt": 0, "add": 0, "decrypt": 0}
    t0 = time.perf_counter()
    for _ in range(n_rounds):
        total = hospitals[0].encrypted_gradient_scalar()
        calls["encrypt"] += len(total)
        for h in hospitals[1:]:
            mine = h.encrypted_gradient_scalar()
            calls["encrypt"] += len(mine)
            total = [a + b for a, b in zip(total, mine)]
            calls["add"] += len(mine)
        mean_gradient = np.array([private_key.decrypt(e) for e in total]) / len(hospitals)
        calls["decrypt"] += len(total)
        for h in hospitals:
            h.step(mean_gradient)
    elapsed = time.perf_counter() - t0
    errors = [h.test_error(X_test, y_test) for h in hospitals]
    if verbose:
        print("federated rounds (scalar API): %d, key: %d bits, %.2f s, calls %s" % (n_rounds, key_length, elapsed, calls))
        for i, e in enumerate(errors, 1):
            print("Hospital %d:\t%.2f" % (i, e))
    return errors, elapsed, calls


def run(key_length=2048, n_rounds=N_ROUNDS, verbose=True, precompute=False):
    """precompute=True: the obfuscators r^n of every encryption of the run are made ahead of time in ONE launch
    (PaillierPublicKey.precompute_obfuscators: the offline half of Paillier encryption); the rounds then only pay the
    online product per gradient entry.  The decrypted aggregates — hence the models — are the same either way."""
    parts, X_test, y_test = load_split(N_HOSPITALS)
    public_key, private_key = paillier.generate_paillier_keypair(n_length=key_length)
    hospitals = [Hospital(X, y, public_key) for X, y in parts]
    if precompute and hasattr(public_key, "precompute_obfuscators") and hasattr(public_key._get_engine().ctx, "encrypt_dev"):
        t_off = time.perf_counter()
        public_key.precompute_obfuscators(n_rounds * len(hospitals) * parts[0][0].shape[1])
        if verbose:
            print("offline: %d obfuscators in %.3f s" % (public_key.obfuscators_available(), time.perf_counter() - t_off))
    t0 = time.perf_counter()
    for _ in range(n_rounds):
        total = hospitals[0].encrypted_gradient()
        for h in hospitals[1:]:
            total = total + h.encrypted_gradient()                 # homomorphic vector add, one launch
        mean_gradient = np.array(private_key.decrypt_batch(total)) / len(hospitals)
        for h in hospitals:
            h.step(mean_gradient)
    elapsed = time.perf_counter() - t0
    errors = [h.test_error(X_test, y_test) for h in hospitals]
    if verbose:
        print("federated rounds: %d, key: %d bits, %.1f s" % (n_rounds, key_length, elapsed))
        for i, e in enumerate(errors, 1):
            print("Hospital %d:\t%.2f" % (i, e))
    return errors, elapsed


def local_only(n_rounds=N_ROUNDS):
    """Each hospital trains alone (no encryption involved): the 'before' numbers of the reference's printout."""
    parts, X_test, y_test = load_split(N_HOSPITALS)
    out = []
    for X, y in parts:
        h = Hospital(X, y, None)
        for _ in range(n_rounds):
            h.step(h.gradient())
        out.append(h.test_error(X_test, y_test))
    return out


if __name__ == "__main__":
    bits = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    print("local-only test MSE:", ", ".join("%.2f" % e for e in local_only()))
    run(bits)
    run(bits, precompute=True)
    run_scalar(bits)
