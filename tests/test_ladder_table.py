"""The committed measured ladder (python-paillier_amd/phe/ladder_gfx950.txt, made by tools/calibrate_ladder.py on an MI355X) is text the
library parses (csrc/phe_hip.hip phe_hip_ctx_load_ladder): every data line must be "key_bits family G rows ns" with a known family and
group width, sizes ascending per rung, and — the point of measuring — every (key width, family) must come with at least two rungs,
the narrowest of which wins at the largest measured size.  GPU: the table is what a fresh context picks its rungs from."""
import os
import sys

import pytest

from conftest import PKG, load_golden

if PKG not in sys.path:
    sys.path.insert(0, PKG)

from phe import _native  # noqa: E402


def _table():
    if not os.path.exists(_native.LADDER_FILE):
        pytest.skip("no committed calibration")
    rows = {}
    with open(_native.LADDER_FILE) as f:
        for line in f:
            line = line.split("#")[0].strip()
            if not line:
                continue
            bits, fam, G, n, ns = line.split()
            rows.setdefault((int(bits), int(fam)), {}).setdefault(int(G), []).append((float(n), float(ns)))
    return rows


def test_committed_table_is_well_formed():
    rows = _table()
    assert {b for b, _ in rows} >= {1024, 2048, 3072} and {f for _, f in rows} == {1, 2, 3}
    for (bits, fam), rungs in rows.items():
        assert len(rungs) >= 2 and set(rungs) <= {1, 2, 4, 8, 16, 64}, (bits, fam, sorted(rungs))
        for G, pts in rungs.items():
            sizes = [n for n, _ in pts]
            assert sizes == sorted(set(sizes)) and all(ns > 0 for _, ns in pts), (bits, fam, G)
        # at the largest size anything was measured at, one of the two narrowest rungs measured there is the fastest
        top = max(n for pts in rungs.values() for n, _ in pts)
        at_top = {G: dict(pts)[top] for G, pts in rungs.items() if top in dict(pts)}
        assert min(at_top, key=at_top.get) in sorted(at_top)[:2], (bits, fam, at_top)


@pytest.mark.gpu
def test_a_fresh_context_takes_its_rungs_from_the_table(monkeypatch):
    import numpy as np
    rows = _table()
    g = load_golden(2048)
    H = lambda k: int(g[k], 16)
    ctx = _native.Context(H("n"), H("p"), H("q"), H("hp"), H("hq"), H("p_inverse"), device=0)
    assert ctx.measured_ladder_lines == sum(len(p) for (b, _), r in rows.items() if b == 2048 for p in r.values()) > 0
    rng = np.random.RandomState(2)
    B = 1 << 11
    m = rng.randint(0, 2 ** 32, size=(B, 64), dtype=np.uint64).astype(np.uint32)
    r = rng.randint(0, 2 ** 32, size=(B, 64), dtype=np.uint64).astype(np.uint32)
    m[:, 63] = 0
    r[:, 63] &= 0x3fffffff
    r[:, 0] |= 1

    def best(fam, n):                      # what the table says: least interpolated time (sizes here are measured sizes)
        t = {G: dict(pts).get(float(n)) for G, pts in rows[(2048, fam)].items()}
        t = {G: v for G, v in t.items() if v is not None}
        return min(t, key=t.get)
    for n in (1 << 8, 1 << 11):
        c = ctx.encrypt(m[:n], r[:n])
        picked = ctx.last_launch()["geom_pub"] // 100
        assert picked == best(1, n), (n, picked)
        back = ctx.decrypt(c)
        assert np.array_equal(back, m[:n])
        assert ctx.last_launch()["geom_priv"] // 100 == best(2, n)
    # a table that makes the whole-wave rung free takes every size there; forgetting the table restores the estimate
    fake = "\n".join("2048 1 %d 1024 %d" % (G, 1 if G == 64 else 10 ** 9) for G in rows[(2048, 1)])
    assert ctx.load_ladder(fake) == len(rows[(2048, 1)])
    c2 = ctx.encrypt(m, r)
    assert ctx.last_launch()["geom_pub"] // 100 == 64
    ctx.load_ladder(None)
    c3 = ctx.encrypt(m, r)
    assert ctx.last_launch()["geom_pub"] // 100 != 64 and np.array_equal(c2, c3)


def test_committed_table_names_the_chip_it_was_measured_on():
    with open(_native.LADDER_FILE) as f:
        head = f.read(600)
    assert "arch: gfx950" in head and "cus: 256" in head


@pytest.mark.gpu
def test_a_table_of_another_chip_or_a_malformed_one_leaves_the_estimate(monkeypatch, tmp_path):
    """launch times of another part must not pick this device's rungs (the header's arch / CU count decide), and
    a malformed table named by PHE_HIP_LADDER_FILE must not stop a key from being created."""
    g = load_golden(2048)
    H = lambda k: int(g[k], 16)
    ctx = _native.Context(H("n"), device=0)
    own = ctx.measured_ladder_lines
    assert own > 0
    body = "2048 1 4 1024 5\n2048 1 16 1024 9\n"
    assert ctx.load_ladder("# arch: gfx950 cus: 256\n" + body) == 2
    assert ctx.load_ladder("# arch: gfx942 cus: 256\n" + body) == 0          # another architecture
    assert ctx.load_ladder("# arch: gfx950 cus: 304\n" + body) == 0          # another CU count
    assert ctx.load_ladder(body) == 2                                        # no header: taken (tests, hand-made tables)
    bad = tmp_path / "bad_ladder.txt"
    bad.write_text("2048 1 4 1024 5\n2048 1 4 1024 6\n")                     # one batch size twice for one rung
    monkeypatch.setattr(_native, "_ladder_text", None)
    monkeypatch.setenv("PHE_HIP_LADDER_FILE", str(bad))
    ctx2 = _native.Context(H("n"), device=0)                                 # does not raise
    assert ctx2.measured_ladder_lines == 0 and "twice" in ctx2.ladder_error
    monkeypatch.setattr(_native, "_ladder_text", None)
