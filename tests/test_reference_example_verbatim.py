"""BASELINE.json configs[4] names a FILE: /root/reference/examples/federated_learning_with_encryption.py (scalar
`public_key.encrypt(x)` loops at :122-133, the protocol at :213-225, `key_length: 1024` at :257).  This test runs that file
verbatim — `runpy.run_path`, nothing copied or patched, the drop-in `phe` first on `sys.path`, the file's own 1024-bit key and 50
rounds — and asserts what the reference itself prints: the five local-only errors and 3775.50 for every hospital after the
protocol.

The build container has no GPU, and the file makes 2,750 one-row encryptions, 550 one-row decryptions and 2,200 additions:
  * default: the hot calls are answered by the libgmp stand-in (tests/host_backend.py) — what is under test is the drop-in's
    Python layer (object API, encoding, exponent alignment, obfuscation flags, engine plumbing) exactly as the file drives it;
  * PHE_TEST_FEDERATED_BACKEND=emu: the same run on the wave emulator (the DEVICE headers compiled for the host, ~40 minutes at
    1024 bits; PHE_TEST_FEDERATED_KEY_BITS=256 lowers the key the file asks for: ~10 minutes).  Recorded:
    profiles/r05_reference_federated_example_verbatim.txt.
The GPU box has no /root/reference: skipped there (tests/test_federated_example.py runs the protocol on the GPU, and
tools/federated_scalar_shape.py times the file's scalar-loop shape there)."""
import os
import re
import runpy
import sys

import pytest

from conftest import PKG

REF_EXAMPLE = "/root/reference/examples/federated_learning_with_encryption.py"

pytestmark = pytest.mark.skipif(not os.path.exists(REF_EXAMPLE), reason="the reference tree is only present in the build container")


def _run_verbatim(monkeypatch, capsys, key_bits):
    monkeypatch.syspath_prepend(PKG)          # the drop-in is what `import phe as paillier` (:74) resolves to
    backend = os.environ.get("PHE_TEST_FEDERATED_BACKEND", "gmp")
    if backend == "emu":
        import emu_backend
        emu_backend.install(monkeypatch)
    else:
        import host_backend
        host_backend.install(monkeypatch)
    import phe
    assert os.path.realpath(os.path.dirname(phe.__file__)).startswith(os.path.realpath(PKG)), phe.__file__
    asked = []
    if key_bits is not None:
        real = phe.generate_paillier_keypair

        def keypair(private_keyring=None, n_length=phe.paillier.DEFAULT_KEYSIZE):
            asked.append(n_length)
            return real(private_keyring, n_length=key_bits)

        monkeypatch.setattr(phe, "generate_paillier_keypair", keypair)
    runpy.run_path(REF_EXAMPLE, run_name="__main__")
    return capsys.readouterr().out, asked


def _errors(block):
    return re.findall(r"Hospital \d+:\s+([0-9.]+)", block)


def test_the_file_configs4_names_runs_verbatim_on_the_drop_in(monkeypatch, capsys):
    key_bits = int(os.environ.get("PHE_TEST_FEDERATED_KEY_BITS", "1024"))
    out, asked = _run_verbatim(monkeypatch, capsys, None if key_bits == 1024 else key_bits)
    if key_bits != 1024:
        assert asked == [1024]            # the example asked for its own key size exactly once (:140, :257)
    local, _, protocol = out.partition("Running distributed gradient aggregation")
    assert _errors(local) == ["3810.44", "3982.58", "3569.32", "4144.15", "3848.39"], out
    assert _errors(protocol) == ["3775.50"] * 5, out
    sys.stderr.write("verbatim reference example, %d-bit key, backend %s:\n%s" % (key_bits, os.environ.get("PHE_TEST_FEDERATED_BACKEND", "gmp"), out))
