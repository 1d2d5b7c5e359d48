"""GPU tests of round 3 (run with -m gpu on an MI355X), all through the C-ABI:
  * the geometry ladder: every rung gives the golden bits, and the rung is picked from the batch size;
  * the paths that only large batches / key owners take (scaled modulus, CRT form of raw_encrypt) on the golden edge
    plaintexts m in {n-1, n, n+1, max_int +- 1} unreduced (phe/paillier.py:134, phe/tests/paillier_test.py:114-126),
    with the path that ran asserted through phe_hip_ctx_last_launch;
  * one stream order per context;
  * resident rows in the pair form (phe_hip_to_pair_dev / pair_mul_dev / from_pair_dev / pair_reduce_dev);
  * the ctypes stub of INTEGRATION.md section B, extracted from the document and executed.
Nothing here reads /root/reference."""
import os
import random
import re
import sys

import numpy as np
import pytest

from conftest import PKG, ROOT, load_golden

if PKG not in sys.path:
    sys.path.insert(0, PKG)

pytestmark = pytest.mark.gpu


def H(x):
    return int(x, 16)


@pytest.fixture(scope="module")
def native():
    from phe import _native
    assert _native.device_count() >= 1
    return _native


def make_ctx(native, g, private=True):
    if private:
        return native.Context(H(g["n"]), H(g["p"]), H(g["q"]), H(g["hp"]), H(g["hq"]), H(g["p_inverse"]),
                              n_limbs=g["key_bits"] // 32)
    return native.Context(H(g["n"]), n_limbs=g["key_bits"] // 32)


def golden_hot_path(native, ctx, g):
    """encrypt / decrypt / obfuscate / powmod of the fixture through ctx; returns what last_launch said after each"""
    key_bits = g["key_bits"]
    s1, s2 = key_bits // 32, key_bits // 16
    L = native.ints_to_limbs
    seen = {}
    enc = g["raw_encrypt"]
    c = ctx.encrypt(L([H(e["m"]) for e in enc], s1), L([H(e["r"]) for e in enc], s1))
    assert native.limbs_to_ints(c) == [H(e["c"]) for e in enc]
    seen["encrypt"] = ctx.last_launch()
    dec = g["raw_decrypt"]
    if ctx.has_private:
        m = ctx.decrypt(L([H(e["c"]) for e in dec], s2))
        assert native.limbs_to_ints(m) == [H(e["m"]) for e in dec]
        seen["decrypt"] = ctx.last_launch()
    obf = g["obfuscate"]
    c2 = ctx.obfuscate(L([H(e["c_in"]) for e in obf], s2), L([H(e["r"]) for e in obf], s1))
    assert native.limbs_to_ints(c2) == [H(e["c_out"]) for e in obf]
    seen["obfuscate"] = ctx.last_launch()
    n_int, max_int = H(g["n"]), H(g["max_int"])
    pos = [e for e in g["raw_mul"] if H(e["s"]) < n_int - max_int]
    out = ctx.powmod(L([H(e["c"]) for e in pos], s2), L([H(e["s"]) for e in pos], s1))
    assert native.limbs_to_ints(out) == [H(e["out"]) for e in pos]
    add = g["raw_add"]
    out = ctx.mulmod(L([H(e["a"]) for e in add], s2), L([H(e["b"]) for e in add], s2))
    assert native.limbs_to_ints(out) == [H(e["out"]) for e in add]
    return seen


@pytest.mark.parametrize("key_bits", [1024, 2048, 3072])
def test_every_rung_of_the_ladder_gives_the_golden_bits(native, key_bits):
    g = load_golden(key_bits)
    ctx = make_ctx(native, g)
    pub, priv = ctx.ladder()
    assert len(pub) >= 2 and len(priv) >= 2, (pub, priv)
    assert pub == sorted(pub, key=lambda c: c // 100) and priv == sorted(priv, key=lambda c: c // 100)   # narrowest first
    lanes = lambda code: (code // 100) * (code % 100)
    assert all(lanes(c) * 29 >= key_bits + 4 for c in pub) and all(lanes(c) * 29 >= key_bits // 2 + 4 for c in priv)
    for width in sorted({c // 100 for c in pub + priv}):
        ctx.set_group(width)
        seen = golden_hot_path(native, ctx, g)
        want_pub = next((c for c in pub if c // 100 >= width), pub[-1])
        want_priv = next((c for c in priv if c // 100 >= width), priv[-1])
        path_unit = seen["encrypt"]["path"] & ctx.PATH_UNIT
        assert seen["encrypt"]["geom_pub"] == want_pub, (width, seen)
        # the scaled modulus k*n rides on every rung whose limbs leave room for its 29 extra bits (rung 0: from 2048-bit keys
        # on), never on the whole-wave rung (whose wave pairs have their own scaled form)
        if want_pub == pub[0]:
            assert bool(path_unit) == (key_bits >= 2048), (width, seen)
        if want_pub // 100 == 64:
            assert not path_unit, (width, seen)
        assert seen["decrypt"]["geom_priv"] == want_priv, (width, seen)
    ctx.set_group(0)


def test_the_rung_follows_the_batch_size(native, c_oracle):
    """2048-bit key: 100 rows -> one number per wavefront, a few thousand -> 16- / 8-lane groups, tens of thousands -> rung 0;
    the decrypt halves run side by side (one grid) while both fit one residency; every size bit-exact (oracle sample + round trip)"""
    g = load_golden(2048)
    n_int = H(g["n"])
    n = native.int_to_limbs(n_int, 64)
    ctx = make_ctx(native, g)
    pub, priv = ctx.ladder()
    rs = np.random.Generator(np.random.PCG64(99))
    widths_pub, widths_priv = [], []
    for batch in (100, 3000, 9000, 20000, 70000):
        m = rs.integers(0, 1 << 32, size=(batch, 64), dtype=np.uint32)
        r = rs.integers(0, 1 << 32, size=(batch, 64), dtype=np.uint32)
        m[:, 63] = 0
        r[:, 63] &= 0x3fffffff
        r[:, 0] |= 1
        c = ctx.encrypt(m, r)
        info = ctx.last_launch()
        widths_pub.append(info["geom_pub"] // 100)
        back = ctx.decrypt(c)
        info = ctx.last_launch()
        widths_priv.append((info["geom_priv"] // 100, bool(info["path"] & ctx.PATH_SIDE_BY_SIDE)))
        assert np.array_equal(back, m), batch
        idx = np.arange(0, batch, max(1, batch // 37))
        assert np.array_equal(c[idx], c_oracle.encrypt(n, m[idx], r[idx], nthreads=8)), batch
    assert widths_pub == sorted(widths_pub, reverse=True) and widths_pub[0] == pub[-1] // 100 and widths_pub[-1] == pub[0] // 100
    assert len(set(widths_pub)) >= 3, widths_pub                     # at least three rungs were exercised
    assert [w for w, _ in widths_priv] == sorted([w for w, _ in widths_priv], reverse=True)
    assert widths_priv[0][1] and not widths_priv[-1][1], widths_priv   # small: halves side by side; 70000 rows: one after the other
    assert widths_priv[-1][0] == priv[0] // 100


@pytest.mark.parametrize("key_bits", [1024, 2048, 3072])
def test_small_batch_rungs_run_the_late_sweeps(native, c_oracle, key_bits, monkeypatch):
    """Round 4: the rungs of 16 lanes and of the whole wave run encrypt / obfuscate / the decrypt halves
    on the LATE sweeps (k_modexp_split_late: scaled modulus, quotient product after the shift, rows = the limbs the modulus
    needs) — the path asserted through last_launch, every golden vector on both rungs, random batches of 100 ... 5000 rows
    against libgmp, the full round trip, and against what the round-3 kernels give (PHE_HIP_NO_LATE)."""
    g = load_golden(key_bits)
    s1, s2 = key_bits // 32, key_bits // 16
    n_int = H(g["n"])
    n = native.int_to_limbs(n_int, s1)
    monkeypatch.setenv("PHE_HIP_NO_WAVE_PAIRS", "1")             # the whole-wave rung on its SINGLE-wave kernels for a handful too
    ctx = make_ctx(native, g)
    for width in (16, 64):
        ctx.set_group(width)
        seen = golden_hot_path(native, ctx, g)
        for op in ("encrypt", "decrypt", "obfuscate"):
            assert seen[op]["path"] & ctx.PATH_LATE, (width, op, seen)
        assert seen["encrypt"]["geom_pub"] // 100 >= width and seen["decrypt"]["geom_priv"] // 100 >= width, (width, seen)   # (the next wider rung where the key has none of this width)
    ctx.set_group(0)
    monkeypatch.setenv("PHE_HIP_NO_LATE", "1")
    plain = make_ctx(native, g)
    monkeypatch.delenv("PHE_HIP_NO_LATE")
    rs = np.random.Generator(np.random.PCG64(key_bits + 4))
    took_late = 0
    for batch in (100, 700, 1100, 2500, 5000):
        m = rs.integers(0, 1 << 32, size=(batch, s1), dtype=np.uint32)
        r = rs.integers(0, 1 << 32, size=(batch, s1), dtype=np.uint32)
        m[:, s1 - 1] = 0
        r[:, s1 - 1] &= 0x3fffffff
        r[:, 0] |= 1
        c = ctx.encrypt(m, r)
        enc_info = ctx.last_launch()
        back = ctx.decrypt(c)
        dec_info = ctx.last_launch()
        took_late += bool(enc_info["path"] & ctx.PATH_LATE) + bool(dec_info["path"] & ctx.PATH_LATE)
        if enc_info["geom_pub"] // 100 >= 16:
            assert enc_info["path"] & ctx.PATH_LATE, (batch, enc_info)
        if dec_info["geom_priv"] // 100 >= 16:
            assert dec_info["path"] & ctx.PATH_LATE, (batch, dec_info)
        assert np.array_equal(back, m), batch
        assert np.array_equal(c, plain.encrypt(m, r)), batch
        assert not plain.last_launch()["path"] & ctx.PATH_LATE
        assert np.array_equal(plain.decrypt(c), m), batch
        idx = np.arange(0, batch, max(1, batch // 29))
        assert np.array_equal(c[idx], c_oracle.encrypt(n, m[idx], r[idx], nthreads=8)), batch
        c2 = ctx.obfuscate(c, r[::-1].copy())
        assert np.array_equal(c2[idx], c_oracle.obfuscate(n, c[idx], r[::-1][idx].copy(), nthreads=8)), batch
    assert took_late >= 6, took_late


@pytest.mark.parametrize("key_bits", [1024, 2048, 3072])
def test_a_handful_of_numbers_runs_on_wave_pairs(native, c_oracle, key_bits, monkeypatch):
    """1 ... 60 numbers: every exponentiation on a PAIR of wavefronts (k_modexp_split_ab: first words on one wave, second words
    one product behind on the other) — asserted through last_launch, bit-exact against the golden vectors and libgmp, and
    equal to what the single-wave kernels give (PHE_HIP_NO_WAVE_PAIRS)"""
    g = load_golden(key_bits)
    s1, s2 = key_bits // 32, key_bits // 16
    n_int = H(g["n"])
    n = native.int_to_limbs(n_int, s1)
    p, q = native.int_to_limbs(H(g["p"]), s1 // 2), native.int_to_limbs(H(g["q"]), s1 // 2)
    ctx = make_ctx(native, g)
    seen = golden_hot_path(native, ctx, g)                       # a few dozen rows per call: the wave-pair path
    assert seen["encrypt"]["path"] & ctx.PATH_WAVE_PAIRS and seen["encrypt"]["geom_pub"] // 100 == 64, seen
    assert seen["decrypt"]["path"] & ctx.PATH_WAVE_PAIRS and seen["decrypt"]["geom_priv"] // 100 == 64, seen
    assert seen["obfuscate"]["path"] & ctx.PATH_WAVE_PAIRS, seen          # golden obfuscate vectors: r^n on wave pairs, then the product
    rng = random.Random(key_bits)
    monkeypatch.setenv("PHE_HIP_NO_WAVE_PAIRS", "1")
    single = make_ctx(native, g)
    for batch in (1, 2, 7, 60):
        m = native.ints_to_limbs([rng.randrange(0, n_int) for _ in range(batch)], s1)
        r = native.ints_to_limbs([rng.randrange(1, n_int) for _ in range(batch)], s1)
        c = ctx.encrypt(m, r)
        assert ctx.last_launch()["path"] & ctx.PATH_WAVE_PAIRS
        assert np.array_equal(c, c_oracle.encrypt(n, m, r, nthreads=4)), batch
        junk = native.ints_to_limbs([rng.randrange(1, n_int * n_int) for _ in range(batch)], s2)
        both = np.concatenate([c, junk])
        got = ctx.decrypt(both)
        assert ctx.last_launch()["path"] & ctx.PATH_WAVE_PAIRS
        assert np.array_equal(got, c_oracle.decrypt(n, p, q, both, nthreads=4)), batch
        assert np.array_equal(got[:batch], m)
        assert np.array_equal(single.encrypt(m, r), c) and np.array_equal(single.decrypt(both), got)
        if ctx.owner_encrypt_offered():                          # the key owner's r^n: its two CRT halves on wave pairs as well
            assert np.array_equal(ctx.encrypt_owner(m, r), c), batch
            path = ctx.last_launch()["path"]
            assert path & ctx.PATH_OWNER and path & ctx.PATH_WAVE_PAIRS, path
    assert not single.last_launch()["path"] & ctx.PATH_WAVE_PAIRS


@pytest.mark.parametrize("key_bits", [1024, 2048])
def test_batch_sizes_either_side_of_every_switch(native, c_oracle, key_bits):
    """The paths change with the batch size — wave pairs up to 256 / 512 numbers, the tail on one wavefront up to 16 per CU, the
    rungs of the ladder, the decrypt halves in one grid or two launches: every size one below, at and one above a switch gives
    the round trip, libgmp's bits on a sample, and the same ciphertexts as the neighbouring sizes' launches for the rows they share."""
    g = load_golden(key_bits)
    s1, s2 = key_bits // 32, key_bits // 16
    n_int = H(g["n"])
    n = native.int_to_limbs(n_int, s1)
    p, q = native.int_to_limbs(H(g["p"]), s1 // 2), native.int_to_limbs(H(g["q"]), s1 // 2)
    ctx = make_ctx(native, g)
    rs = np.random.Generator(np.random.PCG64(key_bits))
    top = 4200
    m = rs.integers(0, 1 << 32, size=(top, s1), dtype=np.uint32)
    r = rs.integers(0, 1 << 32, size=(top, s1), dtype=np.uint32)
    m[:, s1 - 1] = 0
    r[:, s1 - 1] &= 0x3fffffff
    r[:, 0] |= 1
    full = ctx.encrypt(m, r)
    assert np.array_equal(ctx.decrypt(full), m)
    idx = np.arange(0, top, 97)
    assert np.array_equal(full[idx], c_oracle.encrypt(n, m[idx], r[idx], nthreads=8))
    seen = set()
    for batch in (1, 2, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096, 4097):
        c = ctx.encrypt(m[:batch], r[:batch])
        enc_path = ctx.last_launch()
        assert np.array_equal(c, full[:batch]), batch
        back = ctx.decrypt(full[:batch])
        dec_path = ctx.last_launch()
        assert np.array_equal(back, m[:batch]), batch
        seen.add((enc_path["path"], enc_path["geom_pub"], dec_path["path"], dec_path["geom_priv"]))
    assert len(seen) >= 4, seen                                        # the sizes did cross switches
    junk = rs.integers(0, 1 << 32, size=(513, s2), dtype=np.uint32)
    junk[:, s2 - 1] = 0
    for batch in (256, 257, 512, 513):
        assert np.array_equal(ctx.decrypt(junk[:batch]), c_oracle.decrypt(n, p, q, junk[:batch], nthreads=8)), batch


@pytest.mark.parametrize("key_bits", [1024, 2048, 3072])
def test_scalar_multiplication_of_a_handful_runs_on_wave_pairs(native, c_oracle, key_bits, monkeypatch):
    """phe_hip_powmod (host entry: EncryptedNumber.__mul__ one at a time, small lists): each number on a pair of wavefronts with its
    OWN sliding-window schedule, made on the host where the exponents are; a zero exponent in the batch keeps the general kernel.
    libgmp's bits, the golden _raw_mul vectors of the direct branch, and the same as the per-element kernel."""
    g = load_golden(key_bits)
    s1, s2 = key_bits // 32, key_bits // 16
    n_int = H(g["n"])
    n = native.int_to_limbs(n_int, s1)
    ctx = make_ctx(native, g, private=False)
    monkeypatch.setenv("PHE_HIP_NO_WAVE_PAIRS", "1")
    plain = make_ctx(native, g, private=False)
    rng = random.Random(key_bits + 3)
    direct = [e for e in g["raw_mul"] if 0 < H(e["s"]) < n_int - H(g["max_int"])]
    cs = [H(e["c"]) for e in direct] + [rng.randrange(1, n_int * n_int) for _ in range(6)]
    ks = [H(e["s"]) for e in direct] + [1, 2, (1 << 56) - 1, rng.getrandbits(64), rng.randrange(n_int // 3), 3]
    c, k = native.ints_to_limbs(cs, s2), native.ints_to_limbs(ks, s1)
    for batch in (1, len(cs)):
        got = ctx.powmod(c[:batch], k[:batch])
        assert ctx.last_launch()["path"] & ctx.PATH_WAVE_PAIRS, batch
        assert np.array_equal(got, c_oracle.mul(n, c[:batch], k[:batch], nthreads=4)), batch
        assert np.array_equal(got, plain.powmod(c[:batch], k[:batch])), batch
    assert native.limbs_to_ints(ctx.powmod(c, k))[:len(direct)] == [H(e["out"]) for e in direct]
    k0 = k.copy()
    k0[1] = 0                                                        # c^0 = 1: the general kernel
    got = ctx.powmod(c, k0)
    assert np.array_equal(got, c_oracle.mul(n, c, k0, nthreads=4)) and native.limbs_to_ints(got[1:2]) == [1]


@pytest.mark.parametrize("key_bits", [1600, 2100, 2240])
def test_wave_pairs_on_key_sizes_off_the_grid(native, c_oracle, key_bits):
    """Keys whose limb counts are not round (primes of tests/golden/paillier_odd_sizes_primes.json; the key derived as
    phe/paillier.py:224-235 derives it): the last trip of a wave-pair sweep has 1 (2240 bits), 2 (1600) or 3 (2100) steps, a
    compile-time variant each.  A handful of numbers through encrypt, the key owner's encrypt and decrypt: libgmp's bits, and
    the large-batch kernels' (one launch of 3000 rows: another rung of the ladder) on the same rows."""
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "paillier_odd_sizes_primes.json")) as f:
        pq = json.load(f)[str(key_bits)]
    p_int, q_int = int(pq["p"], 16), int(pq["q"], 16)
    n_int = p_int * q_int
    s1 = 2 * ((key_bits + 63) // 64)
    p, q, hp, hq, pinv = c_oracle.private_constants(n_int, p_int, q_int, s1, s1 // 2)
    ctx = native.Context(n_int, p, q, hp, hq, pinv, n_limbs=s1)
    n = native.int_to_limbs(n_int, s1)
    pl, ql = native.int_to_limbs(p_int, s1 // 2), native.int_to_limbs(q_int, s1 // 2)
    rng = random.Random(key_bits)
    ms = [0, 1, n_int - 1] + [rng.randrange(n_int) for _ in range(9)]
    rs = [n_int - 1, 1, 2] + [rng.randrange(1, n_int) for _ in range(9)]
    m, r = native.ints_to_limbs(ms, s1), native.ints_to_limbs(rs, s1)
    c = ctx.encrypt(m, r)
    assert ctx.last_launch()["path"] & ctx.PATH_WAVE_PAIRS
    assert np.array_equal(c, c_oracle.encrypt(n, m, r, nthreads=4))
    if ctx.owner_encrypt_offered():
        assert np.array_equal(ctx.encrypt_owner(m, r), c)
        assert ctx.last_launch()["path"] & ctx.PATH_WAVE_PAIRS
    got = ctx.decrypt(c)
    path = ctx.last_launch()["path"]
    assert path & ctx.PATH_WAVE_PAIRS and path & ctx.PATH_WAVE_TAIL, path
    assert np.array_equal(got, m) and np.array_equal(got, c_oracle.decrypt(n, pl, ql, c, nthreads=4))
    big_m, big_r = np.tile(m, (250, 1)), np.tile(r, (250, 1))        # 3000 rows: a throughput rung
    big = ctx.encrypt(big_m, big_r)
    assert not ctx.last_launch()["path"] & ctx.PATH_WAVE_PAIRS
    assert np.array_equal(big[:len(ms)], c) and np.array_equal(big[-len(ms):], c)
    assert np.array_equal(ctx.decrypt(big), big_m)


@pytest.mark.parametrize("key_bits", [1024, 2048, 3072])
def test_crt_tail_on_one_wavefront_per_ciphertext(native, c_oracle, key_bits, monkeypatch):
    """The L-function / CRT tail runs one ciphertext per WAVEFRONT (k_decrypt_tail_wave; the per-thread tail is
    a 0.4 ms serial chain at 2048 bits): asserted through last_launch, same plaintexts as libgmp and as the per-thread tail
    (PHE_HIP_NO_WAVE_TAIL) on the golden vectors, edge plaintexts and junk ciphertexts, small batches and large."""
    g = load_golden(key_bits)
    s1, s2 = key_bits // 32, key_bits // 16
    n_int = H(g["n"])
    N = n_int * n_int
    n = native.int_to_limbs(n_int, s1)
    p, q = native.int_to_limbs(H(g["p"]), s1 // 2), native.int_to_limbs(H(g["q"]), s1 // 2)
    ctx = make_ctx(native, g)
    monkeypatch.setenv("PHE_HIP_NO_WAVE_TAIL", "1")
    plain = make_ctx(native, g)
    rng = random.Random(key_bits + 1)
    edge = [0, 1, 2, n_int - 1, n_int - 2, H(g["p"]), H(g["q"]), n_int // 2]
    cts = [H(e["c"]) for e in g["raw_decrypt"]] + [(1 + n_int * m) * pow(3, n_int, N) % N for m in edge]
    want = [H(e["m"]) for e in g["raw_decrypt"]] + edge
    cts += [rng.randrange(1, N) for _ in range(40)]                # not ciphertexts of anything: whatever libgmp's arithmetic gives
    p_int, q_int = H(g["p"]), H(g["q"])
    # ... and some that share a factor with n (c^(p-1) mod p^2 = 0: the reference's l_function floors (0 - 1) // p to -1)
    cts += [0, p_int, 3 * p_int, q_int, n_int, 7 * n_int, p_int * p_int, N - p_int]
    c = native.ints_to_limbs(cts, s2)
    for batch in (1, 3, len(cts)):
        got = ctx.decrypt(c[:batch])
        assert ctx.last_launch()["path"] & ctx.PATH_WAVE_TAIL, batch
        assert np.array_equal(got, c_oracle.decrypt(n, p, q, c[:batch], nthreads=4)), batch
        assert np.array_equal(got, plain.decrypt(c[:batch])), batch
        assert not plain.last_launch()["path"] & ctx.PATH_WAVE_TAIL
    assert native.limbs_to_ints(ctx.decrypt(c))[:len(want)] == want
    # large batches take it too from 1024-bit keys up (it is the faster tail at every batch size there); the per-thread kernel
    # stays behind PHE_HIP_NO_WAVE_TAIL / PHE_HIP_WAVE_TAIL_PER_CU and must give the same rows
    big = np.tile(c, (6000 // len(cts) + 1, 1))[:6000]
    got = ctx.decrypt(big)
    assert ctx.last_launch()["path"] & ctx.PATH_WAVE_TAIL
    assert np.array_equal(got[:len(cts)], ctx.decrypt(c))
    assert np.array_equal(got, plain.decrypt(big))
    monkeypatch.delenv("PHE_HIP_NO_WAVE_TAIL")
    monkeypatch.setenv("PHE_HIP_WAVE_TAIL_PER_CU", "1")              # the round-3 rule of thumb, now a knob: small batches only
    few = make_ctx(native, g)
    assert np.array_equal(few.decrypt(big), got)
    assert not few.last_launch()["path"] & ctx.PATH_WAVE_TAIL
    assert np.array_equal(few.decrypt(c[:3]), got[:3])
    assert few.last_launch()["path"] & ctx.PATH_WAVE_TAIL


@pytest.mark.parametrize("key_bits", [2048, 3072])
def test_scaled_modulus_path_on_the_golden_vectors(native, key_bits, monkeypatch):
    """PHE_HIP_FORCE_UNIT: r^n modulo the scaled modulus k*n for small batches too — every golden raw_encrypt / obfuscate
    vector (m = 0, 1, max_int +- 1, n-1, n, n+1 unreduced; r = 1, n-1) through it, and the path asserted"""
    monkeypatch.setenv("PHE_HIP_FORCE_UNIT", "1")
    g = load_golden(key_bits)
    ctx = make_ctx(native, g, private=False)
    seen = golden_hot_path(native, ctx, g)
    assert seen["encrypt"]["path"] & ctx.PATH_UNIT and seen["obfuscate"]["path"] & ctx.PATH_UNIT
    n_int = H(g["n"])
    ms = [H(e["m"]) for e in g["raw_encrypt"]]
    assert any(m == n_int for m in ms) and any(m == n_int + 1 for m in ms) and any(m == n_int - 1 for m in ms)
    monkeypatch.delenv("PHE_HIP_FORCE_UNIT")
    plain = make_ctx(native, g, private=False)
    seen = golden_hot_path(native, plain, g)
    assert not seen["encrypt"]["path"] & ctx.PATH_UNIT                    # a handful of rows: the latency rung, plain modulus


def test_scaled_modulus_and_owner_paths_on_the_reference_trace(native, monkeypatch):
    """every raw_encrypt / obfuscate call the reference's own suites made under a 2048-bit key
    (tests/golden/reference_suite_trace.json.gz, recorded on the real reference) replayed (a) through the scaled-modulus
    path and (b), where the trace holds the key's primes, through the key owner's CRT form with the plaintexts as the
    suite passed them (m >= n included) — the path asserted each time"""
    import gzip
    import json
    from conftest import GOLDEN
    with gzip.open(os.path.join(GOLDEN, "reference_suite_trace.json.gz"), "rb") as f:
        trace = json.loads(f.read().decode())
    keys = [{k: (int(v, 16) if v is not None else None) for k, v in key.items()} for key in trace["keys"]]
    monkeypatch.setenv("PHE_HIP_FORCE_UNIT", "1")
    L = native.ints_to_limbs
    unit_done = owner_done = 0
    for k, key in enumerate(keys):
        n_int = key["n"]
        if n_int.bit_length() < 2048:
            continue
        s1 = (n_int.bit_length() + 31) // 32
        wrap = 1 << (32 * s1)
        enc = [[int(v, 16) for v in row[3:]] for row in trace["ops"] if row[1] == "enc" and row[2] == k]
        obf = [[int(v, 16) for v in row[3:]] for row in trace["ops"] if row[1] == "obf" and row[2] == k]
        if not enc and not obf:
            continue
        ctx = native.Context(n_int, n_limbs=s1)
        if enc:
            got = ctx.encrypt(L([v[0] % wrap for v in enc], s1), L([v[1] for v in enc], s1))
            assert ctx.last_launch()["path"] & ctx.PATH_UNIT
            assert native.limbs_to_ints(got) == [v[2] for v in enc]
            unit_done += len(enc)
        if obf:
            got = ctx.obfuscate(L([v[0] % (n_int * n_int) for v in obf], 2 * s1), L([v[1] for v in obf], s1))
            assert ctx.last_launch()["path"] & ctx.PATH_UNIT
            assert native.limbs_to_ints(got) == [v[2] for v in obf]
            unit_done += len(obf)
        if key["p"] and enc:
            import phe
            priv = phe.PaillierPrivateKey(phe.PaillierPublicKey(n_int), key["p"], key["q"])
            own = native.Context(n_int, priv.p, priv.q, priv.hp, priv.hq, priv.p_inverse, n_limbs=s1)
            if own.owner_encrypt_offered():
                got = own.encrypt_owner(L([v[0] % wrap for v in enc], s1), L([v[1] for v in enc], s1))
                assert own.last_launch()["path"] & own.PATH_OWNER
                assert native.limbs_to_ints(got) == [v[2] for v in enc]
                owner_done += len(enc)
    assert unit_done >= 100 and owner_done >= 20, (unit_done, owner_done)


@pytest.mark.parametrize("key_bits", [1024, 2048, 3072])
def test_key_owner_encryption_with_unreduced_plaintexts(native, key_bits):
    """encrypt_owner on the golden raw_encrypt vectors AS GIVEN — m = n, n + 1 and n - 1 included, not reduced by the
    caller (phe/paillier.py:134 `(n * plaintext + 1) % nsquare` wraps them; phe/tests/paillier_test.py:114-126) — and
    the CRT path asserted"""
    g = load_golden(key_bits)
    ctx = make_ctx(native, g)
    assert ctx.owner_encrypt_offered()
    s1 = key_bits // 32
    enc = g["raw_encrypt"]
    n_int = H(g["n"])
    assert sum(1 for e in enc if H(e["m"]) >= n_int) >= 2
    c = ctx.encrypt_owner(native.ints_to_limbs([H(e["m"]) for e in enc], s1), native.ints_to_limbs([H(e["r"]) for e in enc], s1))
    assert ctx.last_launch()["path"] & ctx.PATH_OWNER
    assert native.limbs_to_ints(c) == [H(e["c"]) for e in enc]
    edge = [n_int - 1, n_int, n_int + 1, (1 << (32 * s1)) - 1, 0, 1]
    rng = random.Random(key_bits)
    r = [rng.randrange(1, n_int) for _ in edge]
    c = ctx.encrypt_owner(native.ints_to_limbs(edge, s1), native.ints_to_limbs(r, s1))
    n2 = n_int * n_int
    assert native.limbs_to_ints(c) == [(1 + n_int * m) % n2 * pow(rr, n_int, n2) % n2 for m, rr in zip(edge, r)]
    assert native.limbs_to_ints(ctx.decrypt(c)) == [m % n_int for m in edge]


def test_one_stream_order_per_context(native, c_oracle):
    """Two decrypt_dev calls on ONE context issued back to back on two different non-blocking streams, no host
    synchronisation in between: the second must wait for the first (they share the context's intermediates and window
    tables).  Both results are checked; without the ordering the first call's x_p / x_q would be overwritten under it."""
    from phe._device import DeviceArray
    g = load_golden(2048)
    n_int = H(g["n"])
    ctx = make_ctx(native, g)
    rng = np.random.Generator(np.random.PCG64(5))
    B = 6000
    plain = []
    cts = []
    for k in range(2):
        m = rng.integers(0, 1 << 32, size=(B, 64), dtype=np.uint32)
        r = rng.integers(0, 1 << 32, size=(B, 64), dtype=np.uint32)
        m[:, 63] = 0
        r[:, 63] &= 0x3fffffff
        r[:, 0] |= 1
        plain.append(m)
        cts.append(DeviceArray.from_host(ctx, ctx.encrypt(m, r)))
    s_a, s_b = ctx.stream_create(), ctx.stream_create()
    outs = [DeviceArray(ctx, B, 64), DeviceArray(ctx, B, 64)]
    for rep in range(3):
        ctx.decrypt_dev(cts[0].ptr, outs[0].ptr, B, s_a)
        ctx.decrypt_dev(cts[1].ptr, outs[1].ptr, B, s_b)
        ctx.sync(s_b)
        ctx.sync(s_a)
        assert np.array_equal(outs[0].to_host(), plain[0]) and np.array_equal(outs[1].to_host(), plain[1]), rep
    ctx.stream_destroy(s_a)
    ctx.stream_destroy(s_b)


# ---- resident rows in the pair form --------------------------------------------------------------------------------------
@pytest.mark.parametrize("key_bits,batch", [(1024, 300), (2048, 40), (2048, 9000), (2048, 40000), (3072, 700)])
def test_pair_form_entry_points(native, c_oracle, key_bits, batch):
    from phe._device import DeviceArray
    g = load_golden(key_bits)
    n_int = H(g["n"])
    n2, s1, s2 = n_int * n_int, key_bits // 32, key_bits // 16
    n = native.int_to_limbs(n_int, s1)
    ctx = make_ctx(native, g, private=False)
    words = ctx.pair_words()
    assert words and 29 * (words // 2) >= key_bits + 4
    rs = np.random.Generator(np.random.PCG64(key_bits + batch))
    a = rs.integers(0, 1 << 32, size=(batch, s2), dtype=np.uint32)
    b = rs.integers(0, 1 << 32, size=(batch, s2), dtype=np.uint32)
    a[:, s2 - 1] = 0
    b[:, s2 - 1] = 0                                               # < n^2 (n^2 has 2*key_bits bits, the top word is not full)
    a[0] = 0
    a[0, 0] = 1
    a[1] = native.int_to_limbs(n2 - 1, s2)
    a[2] = 0xffffffff                                               # above n^2: to_pair takes any value of the width
    da, db = DeviceArray.from_host(ctx, a), DeviceArray.from_host(ctx, b)
    pa, pb = DeviceArray(ctx, batch, words), DeviceArray(ctx, batch, words)
    ctx.to_pair_dev(da.ptr, pa.ptr, batch)
    ctx.to_pair_dev(db.ptr, pb.ptr, batch)
    out = DeviceArray(ctx, batch, s2)
    ctx.from_pair_dev(pa.ptr, None, out.ptr, batch)
    ctx.sync()
    back = out.to_host()
    idx = np.unique(np.concatenate([np.arange(0, min(batch, 4)), np.arange(0, batch, max(1, batch // 53))]))
    assert native.limbs_to_ints(back[idx]) == [v % n2 for v in native.limbs_to_ints(a[idx])]
    prod = DeviceArray(ctx, batch, words)
    ctx.pair_mul_dev(pa.ptr, pb.ptr, False, prod.ptr, batch)
    for _ in range(3):                                              # a chain of additions stays in the pair form
        ctx.pair_mul_dev(prod.ptr, pb.ptr, False, prod.ptr, batch)
    ctx.from_pair_dev(prod.ptr, None, out.ptr, batch)
    ctx.sync()
    got = out.to_host()
    a_red = c_oracle.add(n, a, native.ints_to_limbs([1] * batch, s2), nthreads=8)      # a mod n^2
    want = a_red
    for _ in range(4):
        want = c_oracle.add(n, want, b, nthreads=8)
    assert np.array_equal(got, want)
    # the single row b for every a, and the plaintext factor on the way out
    ctx.pair_mul_dev(pa.ptr, pb.ptr, True, prod.ptr, batch)
    m = rs.integers(0, 1 << 32, size=(batch, s1), dtype=np.uint32)
    m[0] = native.int_to_limbs(n_int, s1)                           # m = n wraps to 0
    dm = DeviceArray.from_host(ctx, m)
    ctx.from_pair_dev(prod.ptr, dm.ptr, out.ptr, batch)
    ctx.sync()
    got = out.to_host()
    b0 = native.limbs_to_ints(b[:1])[0]
    for i in idx.tolist():
        av, mv = native.limbs_to_ints(a[i:i + 1])[0], native.limbs_to_ints(m[i:i + 1])[0]
        assert native.limbs_to_ints(got[i:i + 1])[0] == av * b0 % n2 * (1 + n_int * (mv % n_int)) % n2, i
    # scalar multiplication that stays in the form: a[i]^k[i] (56-bit scalars, some 0 / 1 / 2^56 - 1) against libgmp's powmod
    k = rs.integers(0, 1 << 32, size=(batch, 2), dtype=np.uint32)
    k[:, 1] &= 0x00ffffff
    k[0], k[1], k[2] = 0, (1, 0), (0xffffffff, 0x00ffffff)
    dk = DeviceArray.from_host(ctx, k)
    ctx.pair_powmod_dev(pa.ptr, dk.ptr, 2, 56, prod.ptr, batch)
    ctx.from_pair_dev(prod.ptr, None, out.ptr, batch)
    ctx.sync()
    sc = np.zeros((len(idx), s1), np.uint32)
    sc[:, :2] = k[idx]
    assert np.array_equal(out.to_host()[idx], c_oracle.mul(n, a_red[idx], sc, nthreads=8))
    plain_pow = DeviceArray(ctx, batch, s2)                            # the same through plain residues in and out
    ctx.from_pair_dev(pa.ptr, None, out.ptr, batch)
    ctx.powmod_dev(out.ptr, dk.ptr, 2, 56, plain_pow.ptr, batch)
    ctx.from_pair_dev(prod.ptr, None, out.ptr, batch)
    ctx.sync()
    assert np.array_equal(out.to_host(), plain_pow.to_host())
    # the tree: product of all rows
    root = DeviceArray(ctx, 1, words)
    ctx.pair_reduce_dev(pb.ptr, batch, root.ptr)
    one = DeviceArray(ctx, 1, s2)
    ctx.from_pair_dev(root.ptr, None, one.ptr, 1)
    ctx.sync()
    total = 1
    for v in native.limbs_to_ints(b):
        total = total * v % n2
    assert native.limbs_to_ints(one.to_host())[0] == total


def test_encrypted_vector_in_pair_form(native):
    """EncryptedVector.to_pair(): `+` between resident vectors is one pair product and stays in pair form, sum() is one call,
    obfuscate() multiplies by pooled r^n in pair form; what leaves the vector is what the plain path gives, bit for bit"""
    import phe
    g = load_golden(2048)
    pub = phe.PaillierPublicKey(H(g["n"]))
    priv = phe.PaillierPrivateKey(pub, H(g["p"]), H(g["q"]))
    xs = np.arange(5000, dtype=np.float64) / 8 - 300
    ys = np.arange(5000, dtype=np.int64) * 3 - 7000
    r = [3 + i for i in range(5000)]
    a = pub.encrypt_batch(xs, r_values=r, device=True)
    b = pub.encrypt_batch(ys, r_values=r, device=True)
    plain_sum = (a + b) + b
    pa = a.to_pair()
    assert pa._pair and pa.on_device
    s = (pa + b) + b.to_pair()
    assert s._pair
    assert s.ciphertexts(False) == plain_sum.ciphertexts(False)          # the same canonical residues
    assert not s._pair                                                     # looking at the rows converted them back
    assert priv.decrypt_batch(s) == (xs + 2 * ys).tolist()
    tot_pair, tot_plain = (pa + b).sum(), (a + b).sum()
    assert tot_pair.ciphertext(False) == tot_plain.ciphertext(False) and tot_pair.exponent == tot_plain.exponent
    assert priv.decrypt(tot_pair) == float(np.sum(xs + ys))
    eng = pub._get_engine()
    assert eng.pair_form()
    pub.precompute_obfuscators(6000)
    peek = eng.peek_obfuscators(2)
    v = a.to_pair()
    before = a.ciphertexts(False)[:2]
    v.obfuscate()
    assert v._pair and all(v._obfuscated) and pub.obfuscators_available() == 1000
    n2 = pub.nsquare
    assert v.ciphertexts(False)[:2] == [c * f % n2 for c, f in zip(before, peek)]
    assert priv.decrypt_batch(v) == xs.tolist()
    fresh = pub.encrypt_batch(ys[:1000], device=True)                      # from the pool (pair form) with the plaintext folded in
    assert pub.obfuscators_available() == 0 and priv.decrypt_batch(fresh) == ys[:1000].tolist()
    # multiplication by non-negative scalars stays in the pair form (negative ones take the inverse branch on residues)
    w = (np.arange(5000, dtype=np.int64) % 97) * 12345
    scaled = a.to_pair() * w
    assert scaled._pair
    plain_scaled = a * w
    assert scaled.ciphertexts(False) == plain_scaled.ciphertexts(False)
    assert priv.decrypt_batch(scaled) == (xs * w).tolist()
    signed = a.to_pair() * (w - 500)
    assert not signed._pair and priv.decrypt_batch(signed) == (xs * (w - 500)).tolist()


def test_integration_stub_of_section_b_runs(native, c_oracle):
    """The code a python-paillier maintainer would paste is extracted from the document and
    executed as it stands against plain key objects; encrypt / decrypt through it are checked against the libgmp oracle"""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    section = text[text.index("## B."):]
    block = re.search(r"```python\n(# phe/hip_backend\.py.*?)```", section, re.S).group(1)
    lib_path = os.path.join(PKG, "lib", "libphe_hip.so")
    assert 'ctypes.CDLL("libphe_hip.so")' in block
    block = block.replace('ctypes.CDLL("libphe_hip.so")', "ctypes.CDLL(%r)" % lib_path)   # the document names the soname only
    scope = {}
    exec(compile(block, "INTEGRATION.md:B", "exec"), scope)
    g = load_golden(2048)

    class Pub:
        n = H(g["n"])

    class Priv:
        p, q, hp, hq, p_inverse = (H(g[k]) for k in ("p", "q", "hp", "hq", "p_inverse"))
    hip = scope["HipContext"](Pub, Priv)
    enc = g["raw_encrypt"]
    cts = hip.raw_encrypt_batch([H(e["m"]) % (1 << 2048) for e in enc], [H(e["r"]) for e in enc])
    assert cts == [H(e["c"]) for e in enc]
    rng = random.Random(11)
    ms = [rng.randrange(Pub.n) for _ in range(33)]
    rs = [rng.randrange(1, Pub.n) for _ in range(33)]
    cts = hip.raw_encrypt_batch(ms, rs)
    n = native.int_to_limbs(Pub.n, 64)
    want = c_oracle.encrypt(n, native.ints_to_limbs(ms, 64), native.ints_to_limbs(rs, 64), nthreads=4)
    assert cts == native.limbs_to_ints(want)
    assert hip.raw_decrypt_batch(cts) == ms
    pub_only = scope["HipContext"](Pub)
    assert pub_only.raw_encrypt_batch(ms[:3], rs[:3]) == cts[:3]
    with pytest.raises(ValueError):                                       # status code -> ValueError, as the stub maps it
        pub_only.raw_decrypt_batch(cts[:1])
    if "ciphertext_strings" in section:                                   # the second block (a method for the same class)
        more = re.search(r"```python\n(    def ciphertext_strings.*?)```", section, re.S).group(1)
        ns = dict(scope)
        exec(compile("class _More(HipContext):\n" + more, "INTEGRATION.md:B2", "exec"), ns)
        ext = ns["_More"](Pub)
        assert ext.ciphertext_strings(cts[:5]) == [str(c) for c in cts[:5]]


@pytest.mark.gpu
@pytest.mark.parametrize("key_bits", [1024, 2048])
def test_a_handful_of_host_rows_goes_through_mapped_pinned_memory(native, c_oracle, key_bits, monkeypatch):
    """Host-pointer calls whose operands fit the mapped staging slots (32 KiB each) let the kernels read the operands from and
    write the result to pinned host memory; beyond that, and with PHE_HIP_NO_MAPPED_STAGING, the hipMemcpy staging runs.
    Same bits from both, for every entry point that has the short cut, either side of the slot size, against libgmp."""
    g = load_golden(key_bits)
    s1, s2 = key_bits // 32, key_bits // 16
    n_int, p_int, q_int = H(g["n"]), H(g["p"]), H(g["q"])
    N = n_int * n_int
    n = native.int_to_limbs(n_int, s1)
    p, q = native.int_to_limbs(p_int, s1 // 2), native.int_to_limbs(q_int, s1 // 2)
    fits = 8192 // s2                                       # rows of a ciphertext per slot
    rng = random.Random(7 * key_bits)
    rows = fits + 3
    m = native.ints_to_limbs([rng.randrange(n_int) for _ in range(rows)], s1)
    r = native.ints_to_limbs([rng.randrange(1, n_int) for _ in range(rows)], s1)
    e = native.ints_to_limbs([rng.randrange(1, 1 << 56) for _ in range(rows)], 2)
    fast = make_ctx(native, g)
    monkeypatch.setenv("PHE_HIP_NO_MAPPED_STAGING", "1")
    slow = make_ctx(native, g)
    c_all = slow.encrypt(m, r)
    assert np.array_equal(c_all, c_oracle.encrypt(n, m, r, nthreads=4))
    for batch in (1, 2, fits - 1, fits, fits + 1, rows):
        c = fast.encrypt(m[:batch], r[:batch])
        assert np.array_equal(c, c_all[:batch]), batch
        assert np.array_equal(fast.encrypt_owner(m[:batch], r[:batch]), c), batch
        assert np.array_equal(fast.decrypt(c), m[:batch]), batch
        assert np.array_equal(fast.obfuscate(c, r[:batch][::-1].copy()), slow.obfuscate(c, r[:batch][::-1].copy())), batch
        prod = fast.mulmod(c, c_all[rows - batch:rows])
        assert np.array_equal(prod, slow.mulmod(c, c_all[rows - batch:rows])), batch
        assert np.array_equal(prod, c_oracle.add(n, c, c_all[rows - batch:rows], nthreads=4)), batch
        assert np.array_equal(fast.add_plain(c, m[:batch][::-1].copy()), slow.add_plain(c, m[:batch][::-1].copy())), batch
        pw = fast.powmod(c, e[:batch])
        assert np.array_equal(pw, slow.powmod(c, e[:batch])), batch
        want = [pow(ci, ei, N) for ci, ei in zip(native.limbs_to_ints(c), native.limbs_to_ints(e[:batch]))]
        assert native.limbs_to_ints(pw) == want, batch
    # a failed call (bad argument) in between leaves the next one intact
    with pytest.raises(ValueError):
        fast.mulmod(c_all[:2], c_all[:3])
    assert np.array_equal(fast.mulmod(c_all[:2], c_all[1:3]), slow.mulmod(c_all[:2], c_all[1:3]))
