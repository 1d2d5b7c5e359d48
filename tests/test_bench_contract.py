"""bench.py's bookkeeping, checked without a GPU: the algorithmic MAC32 counts are exactly SURVEY.md 8(d)'s table, the
executed multiply-add model follows the window sizes key_setup.h picks, and the peak is the calibrated half-rate one."""
import os
import sys

from conftest import ROOT

if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_canonical_mac32_counts_match_survey_table():
    assert bench.mac32_counts(1024) == (10023808, 2569600)
    assert bench.mac32_counts(2048) == (79250560, 20056960)
    assert bench.mac32_counts(3072) == (266342976, 67133952)


def test_executed_multiply_adds_model():
    info = {"lane_limbs_pub": 418, "lane_limbs_priv": 218, "engine_pub": "split", "engine_priv": "split"}
    enc, dec = bench.executed_mads(2048, info)
    H = 72
    # 2048 squarings at 4 H^2, ceil(2048/7) + 32 products at 5 H^2 (6-bit windows), entry 4 H^2, exit 10 H^2, nude factor 2 H^2
    assert enc == (4 * 2048 + 5 * (293 + 32) + 4 + 10 + 2) * H * H
    # decrypt: two half-width exponentiations over 1024-bit exponents on H = 36, inputs of 4 chunks (+ one product)
    h = 36
    assert dec == 2 * (4 * 1024 + 5 * (147 + 32) + 4 * 4 + 5 + 10) * h * h
    assert enc < bench.mac32_counts(2048)[0]                   # the split engine issues fewer multiplies than the canonical count
    full = {"lane_limbs_pub": 436, "lane_limbs_priv": 236, "engine_pub": "full", "engine_priv": "full"}
    enc_full, _ = bench.executed_mads(2048, full)
    assert enc_full == (2048 + 325 + 3) * 2 * 144 * 144


def test_peak_is_the_calibrated_half_rate():
    peak, sustained, src = bench.valu_peak_mac32()
    assert abs(peak - 256 * 4 * 64 * 2.4e9 / 4) < 1
    assert sustained is None or 0.7 * peak < sustained < peak
    assert os.path.exists(os.path.join(ROOT, "profiles", src)) if src else True
