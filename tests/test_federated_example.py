"""BASELINE.json configs[4]: the federated-learning experiment on the batched API (examples/federated_learning_batched.py).
CPU: two rounds through the emulator backend must reproduce the plaintext aggregation exactly (to float rounding).
GPU: the full 50 rounds at 2048 bits must print the reference's numbers (3775.50 for every hospital)."""
import os
import sys

import numpy as np
import pytest

from conftest import PKG, ROOT

sys.path.insert(0, os.path.join(ROOT, "examples"))
if PKG not in sys.path:
    sys.path.insert(0, PKG)

import federated_learning_batched as fed  # noqa: E402


def plaintext_rounds(n_rounds):
    parts, X_test, y_test = fed.load_split(fed.N_HOSPITALS)
    hs = [fed.Hospital(X, y, None) for X, y in parts]
    for _ in range(n_rounds):
        g = np.mean([h.gradient() for h in hs], axis=0)
        for h in hs:
            h.step(g)
    return [h.test_error(X_test, y_test) for h in hs]


def test_local_only_numbers_match_reference_printout():
    got = ["%.2f" % e for e in fed.local_only()]
    assert got == ["3810.44", "3982.58", "3569.32", "4144.15", "3848.39"]


def test_two_rounds_on_emulator_match_plaintext(monkeypatch):
    import emu_backend
    emu_backend.install(monkeypatch)
    errors, _ = fed.run(key_length=256, n_rounds=2, verbose=False)
    assert np.allclose(errors, plaintext_rounds(2), rtol=1e-9)


@pytest.mark.gpu
def test_full_protocol_on_gpu():
    errors, elapsed = fed.run(key_length=2048, verbose=False)
    assert ["%.2f" % e for e in errors] == ["3775.50"] * 5
    assert np.allclose(errors, plaintext_rounds(fed.N_ROUNDS), rtol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("key_bits", [1024, 2048])
def test_scalar_api_protocol_on_gpu(key_bits, capsys):
    """BASELINE configs[4] in the call shape of the file it names (2,750 scalar `encrypt`, 2,200 scalar `+`, 550 scalar `decrypt`,
    examples/federated_learning_with_encryption.py:122-133, :213-225) on the real device, at the file's own 1024 bits (:257) and
    at the 2048 bits configs[4] asks for.  The reference file itself cannot be shipped to the GPU box (the reference may not
    travel in any form: DESIGN §1 "configs[4]"); tests/test_reference_example_verbatim.py runs THAT file verbatim against this
    same drop-in in the build container, so the two together pin file -> drop-in API (there) and drop-in API -> MI355X (here)."""
    errors, elapsed, calls = fed.run_scalar(key_length=key_bits, verbose=True)
    out = capsys.readouterr().out
    assert calls == {"encrypt": 2750, "add": 2200, "decrypt": 550}
    assert ["%.2f" % e for e in errors] == ["3775.50"] * 5, out
    assert np.allclose(errors, plaintext_rounds(fed.N_ROUNDS), rtol=1e-9)
    sys.stderr.write("configs[4], scalar API, %d-bit key on the GPU: %.2f s\n%s" % (key_bits, elapsed, out))


def test_scalar_api_protocol_one_round_on_emulator(monkeypatch):
    import emu_backend
    emu_backend.install(monkeypatch)
    errors, _, calls = fed.run_scalar(key_length=256, n_rounds=1, verbose=False)
    assert calls == {"encrypt": 55, "add": 44, "decrypt": 11}
    assert np.allclose(errors, plaintext_rounds(1), rtol=1e-9)
