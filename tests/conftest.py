import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "python-paillier_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")


# Property tests are deterministic: the same examples every run (no example database, no random seed), so that
# "N passed" is reproducible.  Loaded before any test module creates its settings objects, which inherit from it.
try:
    from hypothesis import settings as _hyp_settings
    _hyp_settings.register_profile("repo", derandomize=True, database=None, deadline=None)
    _hyp_settings.load_profile("repo")
except ImportError:                                                # hypothesis is optional on a bare box
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def c_oracle():
    from oracle.paillier_oracle import COracle, build_c_oracle
    build_c_oracle()
    return COracle()


def load_golden(key_bits):
    import gzip
    import json
    path = os.path.join(GOLDEN, "paillier_%d.json" % key_bits)
    if not os.path.exists(path):                     # the wide keys' fixtures are stored compressed
        with gzip.open(path + ".gz", "rb") as f:
            return json.loads(f.read().decode())
    with open(path) as f:
        return json.load(f)


def load_kat():
    import json
    with open(os.path.join(GOLDEN, "reference_kat.json")) as f:
        return json.load(f)
