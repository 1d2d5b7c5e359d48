#!/usr/bin/env python3
"""Scoring plaintext samples against an encrypted logistic-regression model — batched, on the GPU backend.

The protocol of the reference's examples/logistic_regression_encrypted_model.py: Alice trains a classifier, encrypts
its weights and intercept and hands them to Bob; Bob computes, for each of his samples x, the encrypted score
`intercept + sum_i x_i * w_i` (`Bob.encrypted_score`, :170-177 — one `*` and one `+` per feature, per sample) and sends
the scores back; Alice decrypts them and thresholds at 0.  Here Bob's whole evaluation is ONE call:

    scores = encrypted_model.matvec(X1)        # X1 = [X | 1] as a scipy CSR matrix: the intercept rides along as the
                                               # weight of a constant feature

i.e. two launches of the table-lookup multi-exponentiation (the window tables of every encrypted weight once, then one
square-and-multiply ladder per sample over its nonzero features only — the features Bob.encrypted_score walks), and
Alice's side is one `encrypt_batch` and one `decrypt_batch`.  Every score is, bit for bit, the ciphertext the
reference's chain of `*` and `+` gives on the stored entries of [x | 1] (the reference adds the intercept unscaled,
which changes ciphertext bits and exponent but not the decrypted score).

The reference downloads the Enron spam corpus; there is no network here, so the data is a synthetic two-class problem
of the same shape (sparse non-negative counts).  python examples/encrypted_scoring_batched.py [key_length] [samples] [features]
"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-paillier_amd"))

import phe as paillier  # noqa: E402


def make_data(n_samples, n_features, seed=42):
    """Two classes of sparse count vectors (like bag-of-words rows): class-dependent rates on a tenth of the features."""
    rng = np.random.RandomState(seed)
    y = rng.randint(0, 2, n_samples) * 2 - 1
    rates = np.full((2, n_features), 0.05)
    hot = rng.choice(n_features, max(2, n_features // 10), replace=False)
    rates[0, hot[::2]] += 0.6
    rates[1, hot[1::2]] += 0.6
    X = rng.poisson(rates[(y + 1) // 2]).astype(np.float64)
    return X, y


def fit_logistic(X, y, steps=300, lr=0.5, l2=1e-3):
    """Plain gradient descent on the logistic loss (the reference uses sklearn's LogisticRegression)."""
    w, b = np.zeros(X.shape[1]), 0.0
    for _ in range(steps):
        z = np.clip(y * (X @ w + b), -30, 30)
        g = -y / (1.0 + np.exp(z))
        w -= lr * (X.T @ g / len(y) + l2 * w)
        b -= lr * g.mean()
    return w, b


class Alice:
    """Owns the key pair and the model; sees only encrypted scores come back."""

    def __init__(self, key_length):
        self.pubkey, self.privkey = paillier.generate_paillier_keypair(n_length=key_length)

    def fit(self, X, y):
        self.w, self.b = fit_logistic(X, y)

    def encrypt_model(self, device=False):
        return self.pubkey.encrypt_batch(np.append(self.w, self.b), device=device)     # [weights | intercept], one launch

    def decrypt_scores(self, encrypted_scores):
        return np.array(self.privkey.decrypt_batch(encrypted_scores))                   # one launch


class Bob:
    """Holds plaintext samples and the encrypted model; cannot decrypt anything."""

    def __init__(self, encrypted_model):
        self.model = encrypted_model

    def encrypted_evaluate(self, X):
        # like the reference's Bob, only the NONZERO features of a sample enter its score (plus the constant one that
        # carries the intercept): a CSR matrix goes through the table-lookup multi-exponentiation, all samples at once
        X1 = sp.hstack([sp.csr_matrix(X), np.ones((len(X), 1))], format="csr")
        return self.model.matvec(X1)


def run(key_length=1024, n_samples=512, n_features=128, device=None, verbose=True):
    X, y = make_data(2 * n_samples, n_features)
    X_train, y_train, X_test, y_test = X[:n_samples], y[:n_samples], X[n_samples:], y[n_samples:]
    alice = Alice(key_length)
    alice.fit(X_train, y_train)
    clear_scores = X_test @ alice.w + alice.b
    if device is None:
        device = hasattr(alice.pubkey._get_engine().ctx, "encrypt_dev")
    t0 = time.perf_counter()
    bob = Bob(alice.encrypt_model(device=device))
    t1 = time.perf_counter()
    enc_scores = bob.encrypted_evaluate(X_test)
    t2 = time.perf_counter()
    scores = alice.decrypt_scores(enc_scores)
    t3 = time.perf_counter()
    err_clear = float(np.mean(np.sign(clear_scores) != y_test))
    err_enc = float(np.mean(np.sign(scores) != y_test))
    if verbose:
        print("key %d bits, %d samples x %d features" % (key_length, len(X_test), n_features))
        print("Alice encrypts the model:        %.3f s" % (t1 - t0))
        print("Bob scores all samples:          %.3f s  (%.0f encrypted scores/s, %.2e weight-feature products/s)"
              % (t2 - t1, len(X_test) / (t2 - t1), len(X_test) * (n_features + 1) / (t2 - t1)))
        print("Alice decrypts the scores:       %.3f s" % (t3 - t2))
        print("error in the clear %.4f, through the encrypted model %.4f, max |score difference| %.3e"
              % (err_clear, err_enc, float(np.max(np.abs(scores - clear_scores)))))
    return clear_scores, scores, (err_clear, err_enc), (t1 - t0, t2 - t1, t3 - t2)


if __name__ == "__main__":
    args = [int(a) for a in sys.argv[1:]]
    run(*args)
