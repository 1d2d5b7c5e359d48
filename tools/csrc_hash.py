#!/usr/bin/env python3
"""One number for "which kernels is this": sha256 over the device sources (python-paillier_amd/csrc/*) and the C-ABI header.

Every committed count that bench.py quotes in its roofline (profiles/executed_mads_r*.json, hbm_traffic_r*.json) carries the
hash of the sources it was measured on; bench.py emits `"stale": true` and tests/test_profiles_fresh.py fails when the newest
such file was made from other sources than the tree holds (a kernel change must not leave a constant
from an older kernel in the driver's line unnoticed).  No git needed: the GPU box gets a snapshot without .git.

    python tools/csrc_hash.py            # prints the hash of the tree
"""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "python-paillier_amd", "csrc")


def source_files():
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".h", ".hip", ".inc"))]
    return files + [os.path.join(ROOT, "include", "phe_hip.h")]


def csrc_hash():
    h = hashlib.sha256()
    for path in source_files():
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


if __name__ == "__main__":
    print(csrc_hash())
