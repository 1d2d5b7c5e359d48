"""The pair arithmetic of csrc/split_core.h at the limb level (CPU wave emulator): products, squarings and the way
out of the pair form for lazily reduced, almost-normalised operands, worst-case bounds and dense moduli, against
Python integers; plus the integer model that documents the algebra (tools/exp/split_model.py)."""
import os
import random
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "tools", "exp"))

from emu_lib import Emu, from_r29, to_r29  # noqa: E402

MASK = (1 << 29) - 1


@pytest.fixture(scope="module")
def emu():
    return Emu()


def sloppy(limbs, rng):
    """same value, some limbs pushed to the almost-normalised range (< 2^29 + 2^8) by borrowing from the limb above"""
    limbs = list(limbs)
    for k in range(len(limbs) - 1):
        if limbs[k] < 256 and limbs[k + 1] >= 1 and rng.random() < 0.7:
            limbs[k] += 1 << 29
            limbs[k + 1] -= 1
    return limbs


def pair_rows(pairs, H, rng):
    rows = []
    for x0, x1 in pairs:
        rows.append(sloppy(to_r29(x0, H), rng) + sloppy(to_r29(x1, H), rng))
    return np.array(rows, dtype=np.uint32)


def read_pairs(arr, H):
    out = []
    for row in arr:
        assert int(row.max()) < (1 << 29) + 256, "limbs must stay almost-normalised"
        out.append((from_r29(row[:H]), from_r29(row[H:])))
    return out


@pytest.mark.parametrize("G,L,dense", [(16, 1, False), (16, 3, True), (8, 7, False), (8, 9, True), (4, 9, False),
                                       (4, 18, True), (4, 18, False), (2, 18, False), (2, 9, True), (8, 14, False),
                                       (4, 27, True), (4, 27, False), (2, 27, True)])
def test_pair_products_worst_case_operands(emu, G, L, dense):
    H = G * L
    bits = 29 * H - 4                      # the widest modulus this geometry takes: R = 16 * 2^bits
    rng = random.Random(1000 * G + L)
    if dense:
        n = (1 << bits) - 1                # every limb 2^29 - 1: the largest column sums
    else:
        n = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
    limbs32 = (bits + 31) // 32
    n_arr = np.frombuffer(n.to_bytes(4 * limbs32, "little"), dtype=np.uint32).copy()
    assert emu.split_geometry(n_arr)[0] * emu.split_geometry(n_arr)[1] == H
    n2, R = n * n, 1 << (29 * H)
    Rinv = pow(R, -1, n2)
    value = lambda X: (X[0] - n * X[1]) * Rinv % n2
    groups = 64 // G
    worst = (2 * n - 1, 3 * n - 1)
    xs = [worst, (n, n), (0, 0), (1, 0), (2 * n - 1, 0), (0, 3 * n - 1)]
    ys = [worst, (n, 2 * n), worst, (0, 1), (2 * n - 1, 3 * n - 1), (2 * n - 1, 1)]
    while len(xs) < groups:
        xs.append((rng.randrange(2 * n), rng.randrange(3 * n)))
        ys.append((rng.randrange(2 * n), rng.randrange(3 * n)))
    xs, ys = xs[:groups], ys[:groups]
    X, Y = pair_rows(xs, H, rng), pair_rows(ys, H, rng)

    # (lanes wider than 21 limbs make a product out of single sweeps and its second word stays below 3n, not 2n)
    bound1 = 2 * n if L <= 21 else 3 * n
    for Z, x, y in zip(read_pairs(emu.split_pair_op(G, L, n_arr, 0, X, Y), H), xs, ys):
        assert Z[0] < 2 * n and Z[1] < bound1
        assert value(Z) == value(x) * value(y) % n2
    for Z, x in zip(read_pairs(emu.split_pair_op(G, L, n_arr, 1, X), H), xs):
        assert Z[0] < 2 * n and Z[1] < 2 * n
        assert value(Z) == value(x) * value(x) % n2
    # the way out: canonical residues, including X0 = n (u = n, the one case where u + n*t can reach n^2)
    out_limbs = (2 * bits + 31) // 32
    got = emu.split_pair_op(G, L, n_arr, 2, X, out_limbs=out_limbs)
    for row, x in zip(got, xs):
        assert int.from_bytes(row.tobytes(), "little") == value(x)


def test_integer_model_of_the_algebra():
    import split_model
    for bits, h in ((64, 3), (200, 8), (521, 19)):
        assert split_model.check(bits, h, seed=bits)


def _sq_weight(L, k, r):
    """csrc/split_core.h sq_weight, restated: limb k of a lane, row r of the unrolled trip"""
    c = (k - r) % L
    if c == 0 or 2 * c == L:
        return 1
    return 2 if 2 * c < L else 0


@pytest.mark.parametrize("G,L", [(4, 18), (2, 18), (1, 18), (4, 27), (2, 27), (8, 9), (8, 14), (16, 5), (16, 3), (64, 2), (64, 1), (4, 9), (2, 9)])
def test_the_residue_band_of_a_squaring_covers_every_digit_product_once(G, L):
    """Round 5, the combinatorics behind sq_row (the arithmetic itself is pinned by the tests above and by test_emu_core.py): with row
    i = t L + r and limb j = g L + k the selection depends on (k - r) mod L only — the same in every lane g and trip t — and
      * every unordered pair {i, j}, i != j, is taken with total weight 2 over its two orders (row i x limb j, row j x limb i),
      * every diagonal product x_i^2 once,
      * every lane takes L/2 + 1 (even L) or (L + 1)/2 (odd L) limbs in every row,
      * a column, while it stays L rows in one lane (its k + r constant), takes the weight of L products — the full sweep's, so the
        accumulator bounds of the full sweep hold for the symmetric one."""
    H = G * L
    w = lambda i, j: _sq_weight(L, j % L, i % L)          # row i, the lane that holds limb j
    for i in range(H):
        assert w(i, i) == 1
        for j in range(i + 1, H):
            assert w(i, j) + w(j, i) == 2, (i, j)
    per_row = L // 2 + 1 if L % 2 == 0 else (L + 1) // 2
    for r in range(L):
        assert sum(1 for k in range(L) if _sq_weight(L, k, r)) == per_row
    for s in range(L):                                    # a column's stay: k + r = s (mod L) over the L rows of a trip
        assert sum(_sq_weight(L, (s - r) % L, r) for r in range(L)) == L
