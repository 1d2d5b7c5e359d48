"""examples/encrypted_scoring_batched.py (the protocol of the reference's examples/logistic_regression_encrypted_model.py
on the batched API).  CPU: a small instance through the emulator backend must reproduce the plaintext scores and, per
sample, the ciphertext of the reference's per-feature loop (Bob.encrypted_score, :170-177).  GPU: a 1024-bit run."""
import os
import sys

import numpy as np
import pytest

from conftest import PKG, ROOT

sys.path.insert(0, os.path.join(ROOT, "examples"))
if PKG not in sys.path:
    sys.path.insert(0, PKG)

import encrypted_scoring_batched as ex  # noqa: E402


def test_small_instance_on_emulator_matches_the_per_feature_loop(monkeypatch):
    import emu_backend
    emu_backend.install(monkeypatch)
    clear, scores, (err_clear, err_enc), _ = ex.run(key_length=256, n_samples=6, n_features=5, device=False, verbose=False)
    assert np.allclose(scores, clear, rtol=1e-9, atol=1e-9) and err_clear == err_enc
    # the reference's loop on the scalar API, for one sample: intercept + sum_i x_i * w_i over the nonzero features
    alice = ex.Alice(256)
    alice.w, alice.b = np.array([0.5, -1.25, 2.0, 0.0, 3.5]), -0.75
    model = alice.encrypt_model()
    X = np.array([[1.0, 0.0, 2.0, 0.0, 1.0], [0.0, 3.0, 0.0, 1.0, 0.0]])
    out = ex.Bob(model).encrypted_evaluate(X)
    weights = model.to_numbers()
    for r in range(len(X)):
        score = weights[-1] * 1.0
        for i in np.nonzero(X[r])[0]:
            score += float(X[r, i]) * weights[i]
        assert alice.privkey.decrypt(out[r]) == alice.privkey.decrypt(score) == float(X[r] @ alice.w + alice.b)


@pytest.mark.gpu
def test_scoring_on_gpu():
    clear, scores, (err_clear, err_enc), times = ex.run(key_length=1024, n_samples=400, n_features=96, verbose=False)
    assert np.allclose(scores, clear, rtol=1e-9, atol=1e-9) and err_clear == err_enc
