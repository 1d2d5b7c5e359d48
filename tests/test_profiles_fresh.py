"""The committed counts bench.py quotes (profiles/executed_mads_r*.json: multiply-adds a kernel executes;
profiles/hbm_traffic_r*.json: PMC traffic and VALU instructions) must have been made from the device sources the tree holds:
each file carries the sha256 of python-paillier_amd/csrc + include/phe_hip.h (tools/csrc_hash.py), and the NEWEST file of each
kind must match — a kernel change without a recount fails here instead of leaving an older kernel's constant in the driver's
line (bench.py itself says `"stale": true` in that case)."""
import glob
import json
import os
import sys

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from csrc_hash import csrc_hash, source_files  # noqa: E402


def _newest(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    assert files, pattern
    return files[-1]


def test_hash_covers_every_device_source():
    names = {os.path.basename(f) for f in source_files()}
    assert {"split_core.h", "mont_core.h", "mul_io.h", "key_setup.h", "split_kernels.inc", "phe_hip.hip", "phe_hip.h", "wave_gfx950.h"} <= names
    assert len(csrc_hash()) == 64


def test_newest_executed_mads_count_was_made_from_this_tree():
    path = _newest("executed_mads_r*.json")
    with open(path) as f:
        rec = json.load(f)
    assert rec.get("csrc_sha256") == csrc_hash(), (
        "%s was counted on other device sources: rerun `python tools/count_executed_mads.py --out profiles/executed_mads_rNN.json`"
        % os.path.basename(path))


def test_newest_pmc_traffic_was_measured_on_this_tree():
    path = _newest("hbm_traffic_r*.json")
    with open(path) as f:
        rec = json.load(f)
    assert rec.get("csrc_sha256") == csrc_hash(), (
        "%s was measured on other device sources: rerun `TAG=rNN bash tools/gpu_pmc_traffic.sh` on the GPU box and commit its "
        "hbm_traffic.json" % os.path.basename(path))


def test_bench_marks_stale_counts():
    import bench
    newest = os.path.basename(_newest("executed_mads_r*.json"))
    assert bench.committed_count_is_stale(newest) is False
    old = sorted(glob.glob(os.path.join(ROOT, "profiles", "executed_mads_r*.json")))[0]
    if os.path.basename(old) != newest:
        assert bench.committed_count_is_stale(os.path.basename(old)) is True      # (round 2's file carries no hash: stale)
    assert bench.committed_count_is_stale("no_such_file.json") is True
