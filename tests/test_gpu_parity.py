"""GPU parity tests (run with -m gpu on an MI355X).  Everything goes through the C-ABI
(lib/libphe_hip.so via phe._native) and is compared limb-for-limb with
  * tests/golden/*.json  — values produced by the real reference (tests/golden/gen_golden.py),
  * the libgmp oracle (oracle/) on seeded random batches,
and, at sizes the CPU cannot check exhaustively, through size-independent properties
(decrypt∘encrypt = id, homomorphism, inverse*self = 1).
Nothing here reads /root/reference."""
import random
import sys

import numpy as np
import pytest

from conftest import PKG, load_golden, load_kat

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


def test_wave_primitives(native):
    """The DPP group operations mean what csrc/wave_gfx950.h (and the CPU emulator) say they mean."""
    out = native.selftest_prims(0)
    lanes = np.arange(64)
    for G, base in ((16, 0), (8, 258), (4, 450), (2, 642), (64, 834)):
        down = np.where(lanes % G == G - 1, 0, lanes + 1 + 100)
        up = np.where(lanes % G == 0, 0, lanes - 1 + 100)
        bc = (lanes // G) * G + 100
        assert np.array_equal(out[base:base + 64], down), G
        assert np.array_equal(out[base + 64:base + 128], up), G
        assert np.array_equal(out[base + 128:base + 192], bc), G
    assert np.array_equal(out[192:256], (lanes % 3 == 0).astype(np.uint32))
    mask = sum(1 << int(l) for l in lanes if l % 3 == 0)
    assert int(out[256]) | (int(out[257]) << 32) == mask


def test_reference_kat(native):
    k = load_kat()
    ctx = native.Context(k["n"], k["p"], k["q"], k["hp"], k["hq"], k["p_inverse"])
    c = ctx.encrypt(native.ints_to_limbs([k["m"], 1], 1), native.ints_to_limbs([k["r"], 1], 1))
    assert native.limbs_to_ints(c) == [k["c"], k["encrypt_1_r_1"]]      # phe/tests/paillier_test.py:128-149
    assert native.limbs_to_ints(ctx.decrypt(c)) == [k["m"], 1]


@pytest.fixture(params=["auto", "2", "4", "8", "16"])
def group(request, monkeypatch):
    """Limb-group width: 'auto' = 8-lane groups for large batches, 16-lane groups for small ones; '8' / '16'
    force one geometry for every batch size (PHE_HIP_GROUP is read when a context is created)."""
    if request.param == "auto":
        monkeypatch.delenv("PHE_HIP_GROUP", raising=False)
    else:
        monkeypatch.setenv("PHE_HIP_GROUP", request.param)
    return request.param


@pytest.mark.parametrize("key_bits", [256, 1024, 2048, 3072])
def test_golden_vectors(native, key_bits, group):
    """Every golden vector through the default engine: the split-modulus kernels (k_modexp_split /
    k_modexp_var_split) for encrypt, obfuscate, decrypt and powmod, the full-width kernels for mulmod / invert."""
    check_golden(native, key_bits)


@pytest.mark.parametrize("key_bits", [256, 1024, 2048, 3072])
def test_golden_vectors_full_width_engine(native, key_bits, monkeypatch):
    """PHE_HIP_ENGINE=full: the same vectors through k_modexp_uniform / k_modexp_var on n^2, p^2, q^2."""
    monkeypatch.setenv("PHE_HIP_ENGINE", "full")
    ctx = check_golden(native, key_bits)
    assert ctx.info()["engine_pub"] == "full" and ctx.info()["engine_priv"] == "full"


def check_golden(native, key_bits):
    g = load_golden(key_bits)
    s1, s2 = key_bits // 32, key_bits // 16
    n_int = H(g["n"])
    nsq = n_int * n_int
    ctx = make_ctx(native, g)
    L = native.ints_to_limbs
    enc = g["raw_encrypt"]
    c = ctx.encrypt(L([H(e["m"]) for e in enc], s1), L([H(e["r"]) for e in enc], s1))
    assert native.limbs_to_ints(c) == [H(e["c"]) for e in enc]
    dec = g["raw_decrypt"]
    m = ctx.decrypt(L([H(e["c"]) for e in dec], s2))
    assert native.limbs_to_ints(m) == [H(e["m"]) for e in dec]
    obf = g["obfuscate"]
    c2 = ctx.obfuscate(L([H(e["c_in"]) for e in obf], s2), L([H(e["r"]) for e in obf], s1))
    assert native.limbs_to_ints(c2) == [H(e["c_out"]) for e in obf]
    add = g["raw_add"]
    out = ctx.mulmod(L([H(e["a"]) for e in add], s2), L([H(e["b"]) for e in add], s2))
    assert native.limbs_to_ints(out) == [H(e["out"]) for e in add]
    # _raw_mul, both branches, composed exactly like phe/paillier.py:745-751
    max_int = H(g["max_int"])
    pos = [e for e in g["raw_mul"] if H(e["s"]) < n_int - max_int]
    neg = [e for e in g["raw_mul"] if H(e["s"]) >= n_int - max_int]
    out = ctx.powmod(L([H(e["c"]) for e in pos], s2), L([H(e["s"]) for e in pos], s1))
    assert native.limbs_to_ints(out) == [H(e["out"]) for e in pos]
    inv = ctx.invert(L([H(e["c"]) for e in neg], s2))
    assert native.limbs_to_ints(inv) == [pow(H(e["c"]), -1, nsq) for e in neg]
    out = ctx.powmod(inv, L([n_int - H(e["s"]) for e in neg], s1))
    assert native.limbs_to_ints(out) == [H(e["out"]) for e in neg]
    return ctx


@pytest.mark.parametrize("key_bits,batch", [(1024, 300), (2048, 150), (3072, 40)])
def test_random_batches_vs_gmp_oracle(native, c_oracle, key_bits, batch, group):
    g = load_golden(key_bits)
    s1, s2 = key_bits // 32, key_bits // 16
    n_int = H(g["n"])
    n = native.int_to_limbs(n_int, s1)
    p, q = native.int_to_limbs(H(g["p"]), s1 // 2), native.int_to_limbs(H(g["q"]), s1 // 2)
    ctx = make_ctx(native, g)
    rng = random.Random(key_bits)
    m = native.ints_to_limbs([rng.randrange(0, n_int) for _ in range(batch)], s1)
    r = native.ints_to_limbs([rng.randrange(1, n_int) for _ in range(batch)], s1)
    c = ctx.encrypt(m, r)
    assert np.array_equal(c, c_oracle.encrypt(n, m, r, nthreads=8))
    assert np.array_equal(ctx.decrypt(c), m)
    p_int, q_int = H(g["p"]), H(g["q"])
    # not ciphertexts of anything — and a few that share a factor with n (l_function then floors (0 - 1) // p to -1)
    junk = native.ints_to_limbs([0, p_int, 5 * q_int, n_int, n_int * n_int - q_int] +
                                [rng.randrange(1, n_int * n_int) for _ in range(batch - 5)], s2)
    assert np.array_equal(ctx.decrypt(junk), c_oracle.decrypt(n, p, q, junk, nthreads=8))
    big_junk = np.tile(junk, (5000 // batch + 1, 1))                 # beyond the small-batch tail: one ciphertext per thread
    assert np.array_equal(ctx.decrypt(big_junk)[:batch], c_oracle.decrypt(n, p, q, junk, nthreads=8))
    assert np.array_equal(ctx.mulmod(c, junk), c_oracle.add(n, c, junk, nthreads=8))
    scal = native.ints_to_limbs([rng.getrandbits(rng.choice([1, 8, 27, 56, 64, 130])) for _ in range(batch)], s1)
    assert np.array_equal(ctx.powmod(c, scal), c_oracle.mul(n, c, scal, nthreads=8))
    assert np.array_equal(ctx.obfuscate(junk, r), c_oracle.obfuscate(n, junk, r, nthreads=8))
    # adding a plaintext = multiplying by its nude ciphertext 1 + n*m = raw_encrypt(m, r_value=1) (phe/paillier.py:673-675)
    ones = np.zeros_like(r); ones[:, 0] = 1
    assert np.array_equal(ctx.add_plain(c, m), c_oracle.add(n, c, c_oracle.encrypt(n, m, ones, nthreads=8), nthreads=8))


def test_ragged_and_empty_batches(native, c_oracle):
    g = load_golden(1024)
    n_int = H(g["n"])
    n = native.int_to_limbs(n_int, 32)
    ctx = make_ctx(native, g)
    rng = random.Random(5)
    assert ctx.encrypt(np.zeros((0, 32), np.uint32), np.zeros((0, 32), np.uint32)).shape == (0, 64)
    assert ctx.decrypt(np.zeros((0, 64), np.uint32)).shape == (0, 32)
    for batch in (1, 3, 15, 17, 63, 65, 257):
        m = native.ints_to_limbs([rng.randrange(0, n_int) for _ in range(batch)], 32)
        r = native.ints_to_limbs([rng.randrange(1, n_int) for _ in range(batch)], 32)
        c = ctx.encrypt(m, r)
        assert np.array_equal(c, c_oracle.encrypt(n, m, r, nthreads=4)), batch
        assert np.array_equal(ctx.decrypt(c), m), batch


def test_large_batch_properties(native, c_oracle):
    """Config-2 shaped run scaled to the test budget: 2048-bit key, 2^15 elements; full-batch
    round trip + additive homomorphism, strided sample against the libgmp oracle."""
    g = load_golden(2048)
    n_int = H(g["n"])
    n = native.int_to_limbs(n_int, 64)
    ctx = make_ctx(native, g)
    B = 1 << 15
    rs = np.random.Generator(np.random.PCG64(1234))
    m = rs.integers(0, 1 << 32, size=(B, 64), dtype=np.uint32)
    r = rs.integers(0, 1 << 32, size=(B, 64), dtype=np.uint32)
    m[:, 63] = 0      # m < n: top limb cleared (n has 2048 bits)
    r[:, 63] &= 0x3fffffff
    r[:, 0] |= 1      # r != 0
    c = ctx.encrypt(m, r)
    assert np.array_equal(ctx.decrypt(c), m)
    idx = np.arange(0, B, 257)
    assert np.array_equal(c[idx], c_oracle.encrypt(n, m[idx], r[idx], nthreads=8))
    # E(a)*E(b) decrypts to a+b mod n
    half = B // 2
    s = ctx.decrypt(ctx.mulmod(c[:half], c[half:]))
    a_int = native.limbs_to_ints(m[:64]); b_int = native.limbs_to_ints(m[half:half + 64])
    assert native.limbs_to_ints(s[:64]) == [(x + y) % n_int for x, y in zip(a_int, b_int)]
    inv = ctx.invert(c[:1000])
    one = ctx.mulmod(inv, c[:1000])
    expect = np.zeros_like(one); expect[:, 0] = 1
    assert np.array_equal(one, expect)


def test_invert_reports_first_non_unit(native):
    g = load_golden(256)
    ctx = make_ctx(native, g, private=False)
    n_int = H(g["n"])
    vals = [H(e["c"]) for e in g["raw_encrypt"][:6]]
    vals[4] = H(g["p"]) * 12345          # shares a factor with n^2
    with pytest.raises(ZeroDivisionError) as ei:   # phe/util.py:96-102
        ctx.invert(native.ints_to_limbs(vals, 16))
    assert ei.value.bad_index == 4


def test_decrypt_needs_private_key(native):
    g = load_golden(256)
    ctx = make_ctx(native, g, private=False)
    with pytest.raises(ValueError):
        ctx.decrypt(np.zeros((1, 16), np.uint32))


def test_mismatched_private_key_rejected(native):
    g = load_golden(256)
    with pytest.raises(ValueError):   # phe/paillier.py:218-219
        native.Context(H(g["n"]) + 2, H(g["p"]), H(g["q"]), H(g["hp"]), H(g["hq"]), H(g["p_inverse"]), n_limbs=8)


@pytest.mark.parametrize("key_bits", [128, 512, 4096])
def test_other_key_sizes_roundtrip(native, c_oracle, key_bits):
    """Key widths the reference's own tests use (phe/tests/paillier_test.py:47-60) that have no fixture:
    key material from the libgmp oracle's restatement of PaillierPrivateKey.__init__."""
    rng = random.Random(key_bits)

    def prime(bits):
        while True:
            cand = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
            if pow(2, cand - 1, cand) == 1 and pow(3, cand - 1, cand) == 1 and pow(5, cand - 1, cand) == 1:
                return cand
    while True:
        p, q = prime(key_bits // 2), prime(key_bits // 2)
        if p != q and (p * q).bit_length() == key_bits:
            break
    n_int = p * q
    s1 = key_bits // 32
    p, q, hp, hq, pinv = c_oracle.private_constants(n_int, p, q, s1, max(1, s1 // 2))
    ctx = native.Context(n_int, p, q, hp, hq, pinv, n_limbs=s1)
    batch = 24
    m = native.ints_to_limbs([rng.randrange(0, n_int) for _ in range(batch)], s1)
    r = native.ints_to_limbs([rng.randrange(1, n_int) for _ in range(batch)], s1)
    c = ctx.encrypt(m, r)
    assert np.array_equal(c, c_oracle.encrypt(native.int_to_limbs(n_int, s1), m, r, nthreads=8))
    assert np.array_equal(ctx.decrypt(c), m)


def test_obfuscate_composed_and_fused_forms_agree(native, c_oracle, monkeypatch):
    """phe_hip_obfuscate: r^n by the encrypt instantiation followed by one k_mulmod (default) and the fused
    kModeObfuscate instantiation (PHE_HIP_FUSED_OBFUSCATE, and in-place calls) give the oracle's bits"""
    g = load_golden(2048)
    n_int = H(g["n"])
    n = native.int_to_limbs(n_int, 64)
    ctx = make_ctx(native, g, private=False)
    rng = random.Random(77)
    c = native.ints_to_limbs([rng.randrange(1, n_int * n_int) for _ in range(70)], 128)
    r = native.ints_to_limbs([rng.randrange(1, n_int) for _ in range(70)], 64)
    want = c_oracle.obfuscate(n, c, r, nthreads=8)
    assert np.array_equal(ctx.obfuscate(c, r), want)
    monkeypatch.setenv("PHE_HIP_FUSED_OBFUSCATE", "1")
    assert np.array_equal(ctx.obfuscate(c, r), want)


def test_argument_errors_come_back_as_status_codes(native):
    """the C-ABI never throws: bad arguments are PHE_HIP_EINVAL (+ message) -> ValueError in the binding"""
    g = load_golden(1024)
    ctx = make_ctx(native, g, private=False)
    lib = native.lib()
    one = np.zeros((1, 64), np.uint32)
    one[0, 0] = 1
    d = ctx.malloc(4096)
    try:
        # a sign mask without inverted bases; CSR arrays that do not come together; null output
        assert lib.phe_hip_multiexp_rows_dev(ctx._h, d, None, d, d, 1, 32, d, 1, 1, None) == native.EINVAL
        assert b"inverted" in lib.phe_hip_last_error()
        assert lib.phe_hip_multiexp_csr_dev(ctx._h, d, None, 1, d, None, d, None, 1, 32, None, d, 1, None) == native.EINVAL
        assert lib.phe_hip_multiexp_dev(ctx._h, d, d, 1, 32, None, 1, None) == native.EINVAL
        assert lib.phe_hip_to_decimal_dev(ctx._h, d, 0, d, 10, 1, None) == native.EINVAL
        assert lib.phe_hip_decrypt_dev(ctx._h, d, d, 1, None) == native.EINVAL          # public context
        assert b"private" in lib.phe_hip_last_error()
        assert lib.phe_hip_encrypt_dev(None, d, d, d, 1, None) == native.EINVAL
    finally:
        ctx.free(d, 4096)
    with pytest.raises(ValueError):
        ctx.encrypt(np.zeros((2, 32), np.uint32), np.zeros((3, 32), np.uint32))          # batch sizes differ
    with pytest.raises(ValueError):
        native.miller_rabin(np.array([[9]], np.uint32), np.array([[8]], np.uint32))      # base > n - 2
    # zero-length batches are fine everywhere
    assert ctx.multiexp(np.zeros((0, 64), np.uint32), np.zeros((0, 1), np.uint32)).tolist() == one.tolist()
    assert ctx.to_decimal(np.zeros((0, 64), np.uint32)).shape[0] == 0


# ---- element-wise products: staged operand traffic, one-product form, Montgomery debt (csrc/mul_io.h) -----------------
@pytest.mark.parametrize("key_bits,batch", [(256, 70), (1024, 1000), (2048, 5000), (2048, 3), (3072, 300)])
def test_one_product_entry_and_debt_constants(native, key_bits, batch, group):
    """phe_hip_montmul_dev = a*b/R mod n^2 with R = 2^phe_hip_mont_radix_bits; a row-constant R^(d+1) settles d missing
    factors of R.  Against Python integers (canonical residues, so any correct engine agrees bit for bit)."""
    from phe._device import DeviceArray
    g = load_golden(key_bits)
    ctx = make_ctx(native, g, private=False)
    n = H(g["n"])
    N, s2 = n * n, key_bits // 16
    rng = random.Random(key_bits + batch)
    a = [rng.randrange(N) for _ in range(batch)]
    b = [rng.randrange(N) for _ in range(batch)]
    a[0], b[0] = N - 1, N - 1
    a[-1], b[-1] = 0, 1
    R = 1 << ctx.mont_radix_bits()
    assert R >= 16 * N
    Rinv = pow(R, -1, N)
    da, db = (DeviceArray.from_host(ctx, native.ints_to_limbs(v, s2)) for v in (a, b))
    out = DeviceArray(ctx, batch, s2)
    ctx.montmul_dev(da.ptr, db.ptr, False, out.ptr, batch)
    ctx.sync()
    assert native.limbs_to_ints(out.to_host()) == [x * y * Rinv % N for x, y in zip(a, b)]
    for d in (0, 1, 5):
        debt = DeviceArray.from_host(ctx, native.ints_to_limbs([x * pow(Rinv, d, N) % N for x in a], s2))
        const = DeviceArray.from_host(ctx, native.ints_to_limbs([pow(R, d + 1, N)], s2))
        ctx.montmul_dev(debt.ptr, const.ptr, True, out.ptr, batch)
        ctx.sync()
        assert native.limbs_to_ints(out.to_host()) == a
    # the two-product entry on the same operands (the staged kernel where the geometry offers it)
    ctx.mulmod_dev(da.ptr, db.ptr, out.ptr, batch)
    ctx.sync()
    assert native.limbs_to_ints(out.to_host()) == [x * y % N for x, y in zip(a, b)]


def test_raw_add_of_1024_bit_keys_by_tiles_on_eight_waves(native, c_oracle, monkeypatch):
    """Round 5: 1024-bit keys — BASELINE configs[0]'s size, the one the reference publishes figures for — take
    _raw_add (phe/paillier.py:705-719) as one plain product + one fold BY DEFAULT from 16384 rows on: k_mulmod_tile<9, 8> (csrc/mul_tile.h
    TileShape<9, 8>: 72 columns on 8 waves, two workgroups per CU) instead of two Montgomery products.  The path asserted; every row
    against the Montgomery kernels (PHE_HIP_NO_TABLE_MUL) and a sample against Python integers either side of the switch; golden
    raw_add vectors and edge operands (0, 1, n^2 - 1, all-ones rows, operands above n^2); ragged last tile; in place."""
    from phe._device import DeviceArray
    g = load_golden(1024)
    s2 = 64
    n_int = H(g["n"])
    N = n_int * n_int
    ctx = make_ctx(native, g, private=False)
    monkeypatch.setenv("PHE_HIP_NO_TABLE_MUL", "1")
    plain = make_ctx(native, g, private=False)
    monkeypatch.delenv("PHE_HIP_NO_TABLE_MUL")
    rs = np.random.Generator(np.random.PCG64(1024 + 8))
    top = (1 << (32 * s2)) - 1
    edge = [(H(e["a"]), H(e["b"])) for e in g["raw_add"]] + [(0, 5), (1, N - 1), (N - 1, N - 1), (top, top), (top, 1), (N, 7), (N + 1, N + 1)]
    for batch in (16383, 16384, 40001, 140000):
        a = rs.integers(0, 1 << 32, size=(batch, s2), dtype=np.uint32)
        b = rs.integers(0, 1 << 32, size=(batch, s2), dtype=np.uint32)
        a[:, s2 - 1] &= 0x3fffffff
        b[:, s2 - 1] &= 0x3fffffff
        a[:len(edge)] = native.ints_to_limbs([x for x, _ in edge], s2)
        b[:len(edge)] = native.ints_to_limbs([y for _, y in edge], s2)
        a[-1], b[-1] = a[3], b[3]                                       # (the all-ones rows again, in the ragged last tile)
        da, db = DeviceArray.from_host(ctx, a), DeviceArray.from_host(ctx, b)
        out = DeviceArray(ctx, batch, s2)
        ctx.mulmod_dev(da.ptr, db.ptr, out.ptr, batch)
        ctx.sync()
        path = ctx.last_launch()["path"]
        assert bool(path & ctx.PATH_TILE_MUL) == (batch >= 16384) and bool(path & ctx.PATH_TABLE_MUL) == (batch >= 16384), (batch, path)
        got = out.to_host()
        assert native.limbs_to_ints(got[:len(edge)]) == [x * y % N for x, y in edge], batch
        want = plain.mulmod(a, b)                                        # two Montgomery products, every row
        assert not plain.last_launch()["path"] & ctx.PATH_TABLE_MUL
        assert np.array_equal(got, want), batch
        idx = np.arange(len(edge), batch, max(1, batch // 200))
        assert native.limbs_to_ints(got[idx]) == [int(x) * int(y) % N for x, y in zip(native.limbs_to_ints(a[idx]), native.limbs_to_ints(b[idx]))]
        ctx.mulmod_dev(da.ptr, db.ptr, da.ptr, batch)                   # in place: out = a
        ctx.sync()
        assert np.array_equal(da.to_host(), got), batch
    monkeypatch.setenv("PHE_HIP_NO_TILE8", "1")                          # the switch back: the Montgomery kernels at every size
    old = make_ctx(native, g, private=False)
    a, b = a[:20000], b[:20000]
    assert np.array_equal(old.mulmod(a, b), got[:20000]) and not old.last_launch()["path"] & ctx.PATH_TABLE_MUL


@pytest.mark.parametrize("key_bits", [1024, 2048, 3072])
def test_raw_add_by_one_plain_product_and_one_table_fold(native, c_oracle, key_bits, monkeypatch):
    """Round 4: phe_hip_mulmod / _raw_add (phe/paillier.py:705-719 -> phe/util.py:53-64) as ONE plain
    product + ONE fold against the key's table in LDS (csrc/mul_table.h, k_mulmod_table) — the path asserted, every row of
    batches either side of the switch against libgmp, the golden raw_add vectors and edge operands padded to a full batch,
    in place, and equal to what the two-Montgomery-product kernels give (PHE_HIP_NO_TABLE_MUL); rows off a 16-byte boundary
    and small batches keep those kernels.  (1024-bit keys: the 5-limb form is compiled but not offered by default — slower than
    the Montgomery kernels there — PHE_HIP_TABLE_MUL_ANY_WIDTH offers it.)"""
    from phe._device import DeviceArray
    monkeypatch.setenv("PHE_HIP_TABLE_MUL_ANY_WIDTH", "1")
    monkeypatch.setenv("PHE_HIP_NO_TILE8", "1")     # (1024 bits: the 16 x 5 forms under test here; the 8-wave tiles it takes by default: next test)
    g = load_golden(key_bits)
    s1, s2 = key_bits // 32, key_bits // 16
    n_int = H(g["n"])
    N = n_int * n_int
    n = native.int_to_limbs(n_int, s1)
    ctx = make_ctx(native, g, private=False)
    monkeypatch.setenv("PHE_HIP_NO_TABLE_MUL", "1")
    plain = make_ctx(native, g, private=False)
    monkeypatch.delenv("PHE_HIP_NO_TABLE_MUL")
    monkeypatch.setenv("PHE_HIP_NO_TILE_MUL", "1")                    # the round-4 first form: the table in LDS (mul_table.h)
    in_lds = make_ctx(native, g, private=False)
    monkeypatch.delenv("PHE_HIP_NO_TILE_MUL")
    rs = np.random.Generator(np.random.PCG64(key_bits + 11))
    top = (1 << (32 * s2)) - 1
    edge = [(H(e["a"]), H(e["b"])) for e in g["raw_add"]] + [(0, 5), (1, N - 1), (N - 1, N - 1), (top, top), (top, 1), (N, 7), (N + 1, N + 1)]
    for batch in (8191, 8192, 20000, 70000):
        a = rs.integers(0, 1 << 32, size=(batch, s2), dtype=np.uint32)
        b = rs.integers(0, 1 << 32, size=(batch, s2), dtype=np.uint32)
        a[:, s2 - 1] &= 0x3fffffff                                    # residues below n^2 (what the reference passes); the edge rows
        b[:, s2 - 1] &= 0x3fffffff                                    # below also hold wider operands
        a[:len(edge)] = native.ints_to_limbs([x for x, _ in edge], s2)
        b[:len(edge)] = native.ints_to_limbs([y for _, y in edge], s2)
        da, db = DeviceArray.from_host(ctx, a), DeviceArray.from_host(ctx, b)
        out = DeviceArray(ctx, batch, s2)
        ctx.mulmod_dev(da.ptr, db.ptr, out.ptr, batch)
        ctx.sync()
        took_table = bool(ctx.last_launch()["path"] & ctx.PATH_TABLE_MUL)
        # 3072-bit keys: the table (212 rows of 224 limbs) does not fit LDS — no k_mulmod_table, the tile kernel from 16384 rows on
        assert took_table == (batch >= (8192 if key_bits < 3072 else 16384)), (batch, ctx.last_launch())
        assert bool(ctx.last_launch()["path"] & ctx.PATH_TILE_MUL) == (batch >= 16384)   # by tiles of 64, one element per lane (mul_tile.h)
        got = out.to_host()
        if took_table:
            other = in_lds.mulmod(a, b)
            assert in_lds.last_launch()["path"] & (ctx.PATH_TABLE_MUL | ctx.PATH_TILE_MUL) == (ctx.PATH_TABLE_MUL if key_bits < 3072 else 0)
            assert np.array_equal(got, other), batch
        assert native.limbs_to_ints(got[:len(edge)]) == [x * y % N for x, y in edge], batch
        if batch <= 20000:
            want = plain.mulmod(a[len(edge):], b[len(edge):])                                        # two Montgomery products
            assert not plain.last_launch()["path"] & ctx.PATH_TABLE_MUL
            assert np.array_equal(got[len(edge):], want), batch
        idx = np.arange(len(edge), batch, max(1, batch // 300))
        res = [int(x) * int(y) % N for x, y in zip(native.limbs_to_ints(a[idx]), native.limbs_to_ints(b[idx]))]
        assert native.limbs_to_ints(got[idx]) == res, batch
        ctx.mulmod_dev(da.ptr, db.ptr, da.ptr, batch)                                                  # in place: out = a
        ctx.sync()
        assert np.array_equal(da.to_host(), got), batch
    # host-pointer entry point: same path, same bits
    a = rs.integers(0, 1 << 32, size=(9000, s2), dtype=np.uint32)
    b = rs.integers(0, 1 << 32, size=(9000, s2), dtype=np.uint32)
    got = ctx.mulmod(a, b)
    assert bool(ctx.last_launch()["path"] & ctx.PATH_TABLE_MUL) == (key_bits < 3072)          # (9000 rows)
    assert native.limbs_to_ints(got[:200]) == [int(x) * int(y) % N for x, y in zip(native.limbs_to_ints(a[:200]), native.limbs_to_ints(b[:200]))]
    # rows 4 bytes off a 16-byte boundary: the plain Montgomery body
    flat = lambda v: np.concatenate([np.zeros(1, np.uint32), v.reshape(-1)])
    da = DeviceArray.from_host(ctx, flat(a).reshape(-1, 1))
    db = DeviceArray.from_host(ctx, flat(b).reshape(-1, 1))
    out = DeviceArray(ctx, 9000 * s2 + 1, 1)
    ctx.mulmod_dev(da.ptr + 4, db.ptr + 4, out.ptr + 4, 9000)
    ctx.sync()
    assert not ctx.last_launch()["path"] & ctx.PATH_TABLE_MUL
    assert np.array_equal(out.to_host().reshape(-1)[1:].reshape(9000, s2), got)


@pytest.mark.parametrize("key_bits", [1920, 3040])
def test_raw_add_by_tiles_on_keys_off_the_limb_grid(native, key_bits):
    """The tile kernel (csrc/mul_tile.h) takes any n^2 whose limbs fill the lanes to within 16: keys between the golden sizes —
    1920 bits (n^2: 133 limbs of the 144 of L = 9, 11 fold digits from the low half) and 3040 bits (210 of 224, L = 14) — made
    here with the package's own key generation; every row of a batch on the tile path against Python integers, edge operands in
    front, the path asserted."""
    from phe import paillier
    from phe._device import DeviceArray
    pub, _ = paillier.generate_paillier_keypair(n_length=key_bits)
    n_int = pub.n
    assert n_int.bit_length() == key_bits
    N = n_int * n_int
    s1 = -(-key_bits // 128) * 4                       # words of n, rows of whole 16-byte pieces
    s2 = 2 * s1
    ctx = native.Context(n_int, n_limbs=s1)
    rs = np.random.Generator(np.random.PCG64(key_bits))
    batch = 16384 + 77
    top = (1 << (32 * s2)) - 1
    edge = [(0, 5), (1, N - 1), (N - 1, N - 1), (top, top), (top, 1), (N, 7), (N + 1, N + 1)]
    a = rs.integers(0, 1 << 32, size=(batch, s2), dtype=np.uint32)
    b = rs.integers(0, 1 << 32, size=(batch, s2), dtype=np.uint32)
    a[:len(edge)] = native.ints_to_limbs([x for x, _ in edge], s2)
    b[:len(edge)] = native.ints_to_limbs([y for _, y in edge], s2)
    da, db = DeviceArray.from_host(ctx, a), DeviceArray.from_host(ctx, b)
    out = DeviceArray(ctx, batch, s2)
    ctx.mulmod_dev(da.ptr, db.ptr, out.ptr, batch)
    ctx.sync()
    assert ctx.last_launch()["path"] & ctx.PATH_TILE_MUL, ctx.last_launch()
    got = native.limbs_to_ints(out.to_host())
    idx = list(range(len(edge))) + list(range(len(edge), batch, 37)) + [batch - 1]
    ai, bi = native.limbs_to_ints(a[idx]), native.limbs_to_ints(b[idx])
    assert [got[i] for i in idx] == [int(x) * int(y) % N for x, y in zip(ai, bi)]


def test_negative_scalars_invert_only_the_rows_that_take_the_branch(native, c_oracle):
    """_raw_mul's negative branch (phe/paillier.py:745-749: powmod(invert(c), n - s)) on a resident vector with few negative
    scalars: only those rows are inverted — phe_hip_gather_rows_dev, the simultaneous inversion of the subset,
    phe_hip_scatter_rows_dev into a copy (Engine._inverted_where) — same bits as the whole-vector form and as libgmp; a row
    without an inverse is reported by its index in the VECTOR."""
    from phe._device import DeviceArray
    from phe._engine import Engine
    g = load_golden(2048)
    n_int, s1, s2 = H(g["n"]), 64, 128
    N = n_int * n_int
    eng = Engine(n_int)
    rng = random.Random(77)
    rows = 3000
    cs = [rng.randrange(1, N) for _ in range(rows)]
    mag = np.array([rng.randrange(1, 1 << 50) for _ in range(rows)], dtype=np.uint64)
    neg = np.zeros(rows, dtype=bool)
    neg[rng.sample(range(rows), 211)] = True                       # 7 %: the subset form
    c_dev = eng.upload_cipher(cs)
    src = c_dev.to_host().copy()
    out = eng.raw_mul_signed_dev(c_dev, mag, neg).to_host()
    assert np.array_equal(c_dev.to_host(), src)                    # the operand is left alone (the inverses land in a copy)
    want = [pow(pow(c, -1, N) if ng else c, int(k), N) for c, k, ng in zip(cs, mag.tolist(), neg.tolist())]
    assert native.limbs_to_ints(out) == want
    dense = neg.copy()
    dense[::2] = True                                              # more than a quarter negative: whole-vector inversion + select
    out2 = eng.raw_mul_signed_dev(c_dev, mag, dense).to_host()
    pick = np.nonzero(neg)[0][:40]
    assert np.array_equal(out2[pick], out[pick])
    # the two data movements on their own
    idx = np.array([5, 2999, 0, 77], dtype=np.uint32)
    idx_d = DeviceArray.from_host(eng.ctx, idx)
    sub = DeviceArray(eng.ctx, 4, s2)
    eng.ctx.gather_rows_dev(c_dev.ptr, rows, idx_d.ptr, sub.ptr, s2, 4)
    eng.ctx.sync()
    assert np.array_equal(sub.to_host(), src[idx.astype(np.int64)])
    dst = DeviceArray.from_host(eng.ctx, np.zeros((rows, s2), np.uint32))
    eng.ctx.scatter_rows_dev(sub.ptr, idx_d.ptr, dst.ptr, rows, s2, 4)
    eng.ctx.sync()
    back = dst.to_host()
    assert np.array_equal(back[idx.astype(np.int64)], src[idx.astype(np.int64)]) and int(np.count_nonzero(back.any(axis=1))) == 4
    # an index beyond the indexed buffer never leaves it: the gather gives a row of zeros, the scatter skips the row
    short = 100                                                    # pretend the indexed side holds only 100 rows: 2999 is out of range
    eng.ctx.gather_rows_dev(c_dev.ptr, short, idx_d.ptr, sub.ptr, s2, 4)
    eng.ctx.sync()
    got = sub.to_host()
    assert np.array_equal(got[[0, 2, 3]], src[[5, 0, 77]]) and not got[1].any()
    dst2 = DeviceArray.from_host(eng.ctx, np.zeros((rows, s2), np.uint32))
    sub = DeviceArray.from_host(eng.ctx, src[idx.astype(np.int64)])
    eng.ctx.scatter_rows_dev(sub.ptr, idx_d.ptr, dst2.ptr, short, s2, 4)
    eng.ctx.sync()
    back2 = dst2.to_host()
    assert not back2[2999].any() and int(np.count_nonzero(back2.any(axis=1))) == 3
    # no inverse: the index is the row's place in the vector, not in the subset
    bad = list(cs)
    where = int(np.nonzero(neg)[0][17])
    bad[where] = H(g["p"]) * 12345
    with pytest.raises(ZeroDivisionError) as info:
        eng.raw_mul_signed_dev(eng.upload_cipher(bad), mag, neg)
    assert info.value.bad_index == where


def test_products_on_rows_that_are_not_16_byte_aligned(native, c_oracle):
    """rows that start 4 bytes off a 16-byte boundary take the plain body (no 16-byte chunks): same results"""
    from phe._device import DeviceArray
    g = load_golden(2048)
    ctx = make_ctx(native, g, private=False)
    n = H(g["n"])
    N, s2, batch = n * n, 128, 600
    rng = random.Random(5)
    a = [rng.randrange(N) for _ in range(batch)]
    b = [rng.randrange(N) for _ in range(batch)]
    flat = lambda v: np.concatenate([np.zeros(1, np.uint32), native.ints_to_limbs(v, s2).reshape(-1)])
    da = DeviceArray.from_host(ctx, flat(a).reshape(-1, 1))
    db = DeviceArray.from_host(ctx, flat(b).reshape(-1, 1))
    out = DeviceArray(ctx, batch * s2 + 1, 1)
    R = 1 << ctx.mont_radix_bits()
    Rinv = pow(R, -1, N)
    ctx.mulmod_dev(da.ptr + 4, db.ptr + 4, out.ptr + 4, batch)
    ctx.sync()
    assert native.limbs_to_ints(out.to_host().reshape(-1)[1:].reshape(batch, s2)) == [x * y % N for x, y in zip(a, b)]
    ctx.montmul_dev(da.ptr + 4, db.ptr + 4, False, out.ptr + 4, batch)
    ctx.sync()
    assert native.limbs_to_ints(out.to_host().reshape(-1)[1:].reshape(batch, s2)) == [x * y * Rinv % N for x, y in zip(a, b)]


@pytest.mark.parametrize("count", [1, 2, 7, 1000, 4099])
def test_resident_vectors_add_lazily_and_settle_to_the_reference_bits(native, count):
    """EncryptedVector on the device: `+` is one Montgomery product and leaves a debt, sum() is a tree of them; every way
    of looking at the result (ciphertexts, host copy, decrypt, a further scalar multiplication) gives the bits that the
    chain of _raw_add (phe/paillier.py:705-719) gives on integers"""
    from phe import paillier
    g = load_golden(1024)
    pub = paillier.PaillierPublicKey(H(g["n"]))
    priv = paillier.PaillierPrivateKey(pub, H(g["p"]), H(g["q"]))
    N = pub.nsquare
    rng = random.Random(count)
    xs = [rng.randrange(-10 ** 6, 10 ** 6) for _ in range(count)]
    ys = [rng.randrange(-10 ** 6, 10 ** 6) for _ in range(count)]
    rs = [rng.randrange(1, pub.n) for _ in range(2 * count)]
    va = pub.encrypt_batch(xs, r_values=rs[:count], device=True)
    vb = pub.encrypt_batch(ys, r_values=rs[count:], device=True)
    ca, cb = va.ciphertexts(False), vb.ciphertexts(False)
    s1 = va + vb
    assert s1._debt == 1
    s3 = (s1 + va) + (vb + vb)                       # debts 1 + 0 + 1 = 2, (0 + 0 + 1) = 1 -> 2 + 1 + 1 = 4
    assert s3._debt == 4
    want3 = [a * b % N * a % N * (b * b % N) % N for a, b in zip(ca, cb)]
    total = s3.sum()                                 # the tree runs on the indebted rows
    assert s3._debt == 4
    acc = 1
    for w in want3:
        acc = acc * w % N
    assert total.ciphertext(False) == acc
    assert priv.decrypt(total) == sum(2 * x + 3 * y for x, y in zip(xs, ys))
    assert s3.ciphertexts(False) == want3 and s3._debt == 0
    assert s1.to_host().ciphertexts(False) == [a * b % N for a, b in zip(ca, cb)]
    assert priv.decrypt_batch(va + vb + va) == [2 * x + y for x, y in zip(xs, ys)]
    scaled = (va + vb) * 3                           # a scalar multiplication settles first
    assert scaled.ciphertexts(False) == [pow(a * b % N, 3, N) for a, b in zip(ca, cb)]
    assert va.sum().ciphertext(False) == __import__("functools").reduce(lambda p, c: p * c % N, ca, 1)


def test_large_host_batches_are_pipelined_and_equal_the_blocking_path(native, c_oracle, monkeypatch):
    """host-pointer entry points from two chunks of 65536 rows on: chunks through pinned staging on three streams
    (csrc/phe_hip.hip run_pipelined).  Ragged last chunk; same bits as the blocking path and as the oracle."""
    g = load_golden(1024)
    n = H(g["n"])
    batch, s1 = 2 * 65536 + 4097, 32
    rng = np.random.Generator(np.random.PCG64(11))
    m = rng.integers(0, 2 ** 32, (batch, s1), dtype=np.uint64).astype(np.uint32)
    r = rng.integers(0, 2 ** 32, (batch, s1), dtype=np.uint64).astype(np.uint32)
    m[:, -1] = 0
    r[:, -1] &= 0x3fffffff
    r[:, 0] |= 1
    ctx = make_ctx(native, g)
    c = ctx.encrypt(m, r)                                        # pipelined
    assert np.array_equal(ctx.decrypt(c), m)                      # pipelined decrypt
    idx = np.r_[0, 65535, 65536, 131071, 131072, batch - 1, np.arange(7, batch, 9973)]
    assert np.array_equal(c[idx], c_oracle.encrypt(native.int_to_limbs(n, s1), m[idx], r[idx], nthreads=4))
    c2 = np.ascontiguousarray(c[::-1])
    prod = ctx.mulmod(c, c2)
    assert np.array_equal(prod[idx], c_oracle.add(native.int_to_limbs(n, s1), c[idx], c2[idx], nthreads=4))
    plain = ctx.add_plain(c, m)
    obf = ctx.obfuscate(c, r)
    monkeypatch.setenv("PHE_HIP_NO_PIPELINE", "1")
    ctx2 = make_ctx(native, g)
    assert np.array_equal(ctx2.encrypt(m, r), c)
    assert np.array_equal(ctx2.mulmod(c, c2), prod)
    assert np.array_equal(ctx2.add_plain(c, m), plain)
    assert np.array_equal(ctx2.obfuscate(c, r), obf)


def test_library_rccl_allgather_single_rank(native):
    """phe_hip_comm_* / phe_hip_allgather_dev with a world of one (all a 1-GPU box offers): RCCL loads, the communicator
    comes up on the context's device and the gather is the identity.  N > 1 runs under bench.py --lib-allgather."""
    from phe._device import DeviceArray
    g = load_golden(1024)
    ctx = make_ctx(native, g, private=False)
    comm = native.Communicator(ctx, native.comm_unique_id(), 0, 1)
    rows, limbs = 1000, 64
    data = np.random.Generator(np.random.PCG64(2)).integers(0, 2 ** 32, (rows, limbs), dtype=np.uint64).astype(np.uint32)
    src = DeviceArray.from_host(ctx, data)
    dst = DeviceArray(ctx, rows, limbs)
    comm.allgather_dev(src.ptr, dst.ptr, rows, limbs)
    ctx.sync()
    assert np.array_equal(dst.to_host(), data)
    comm.close()


def test_wide_keys_8192_bits(native, c_oracle):
    """8192-bit keys — what the reference's own benchmark goes up to (examples/benchmarks.py:88-90): n^2 has no full-width
    geometry, so the products (add, add a plaintext, the inversion tree) run on the pair form (k_mulmod_split) next to
    the split-modulus exponentiations.  Golden vectors of the real reference, then a random batch against libgmp."""
    ctx = check_golden(native, 8192)
    info = ctx.info()
    assert info["engine_pub"] == "split" and info["lane_limbs_pub"] // 100 == 16
    g = load_golden(8192)
    n_int, s1, s2 = H(g["n"]), 256, 512
    rng = random.Random(8)
    batch = 48
    m = native.ints_to_limbs([rng.randrange(0, n_int) for _ in range(batch)], s1)
    r = native.ints_to_limbs([rng.randrange(1, n_int) for _ in range(batch)], s1)
    n_arr = native.int_to_limbs(n_int, s1)
    c = ctx.encrypt(m, r)
    assert np.array_equal(c, c_oracle.encrypt(n_arr, m, r, nthreads=8))
    assert np.array_equal(ctx.decrypt(c), m)
    c2 = np.ascontiguousarray(c[::-1])
    assert np.array_equal(ctx.mulmod(c, c2), c_oracle.add(n_arr, c, c2, nthreads=8))
    ones = np.zeros_like(m)
    ones[:, 0] = 1
    assert np.array_equal(ctx.add_plain(c, m), c_oracle.add(n_arr, c, c_oracle.encrypt(n_arr, m, ones, nthreads=8), nthreads=8))
    inv = ctx.invert(c)
    assert np.array_equal(ctx.mulmod(inv, c)[:, 0], np.ones(batch, np.uint32)) and not ctx.mulmod(inv, c)[:, 1:].any()
    with pytest.raises(ValueError):
        ctx.mont_radix_bits()                                    # the one-product form needs the full-width geometry


def test_wide_keys_through_the_drop_in_api():
    """the reference's benchmark loop (examples/benchmarks.py:38-71: encrypt, decrypt, add, multiply by a scalar) on an
    8192-bit key through the drop-in classes, scalar and batched"""
    from phe import paillier
    g = load_golden(8192)
    pub = paillier.PaillierPublicKey(H(g["n"]))
    priv = paillier.PaillierPrivateKey(pub, H(g["p"]), H(g["q"]))
    for e in g["encrypt_api"]:
        value = eval(e["value"])
        en = pub.encrypt(value, r_value=H(e["r"]))
        assert en.ciphertext(False) == H(e["c"]) and en.exponent == e["exponent"]
        assert repr(priv.decrypt(en)) == e["decrypted"]
    xs = [0.5, -3.25, 1e6, 7.0]
    vec = pub.encrypt_batch(np.array(xs), device=True)
    assert priv.decrypt_batch(vec + vec) == [2 * x for x in xs]
    assert priv.decrypt_batch(vec * np.array([2.0, -1.0, 0.5, 3.0])) == [1.0, 3.25, 5e5, 21.0]
    assert priv.decrypt(vec.sum()) == sum(xs)
    nums = [pub.encrypt(x) for x in xs]
    assert priv.decrypt(nums[0] + nums[1] * -2) == 0.5 + 6.5


@pytest.mark.parametrize("key_bits,batch", [(256, 100), (1024, 3000), (2048, 40000), (3072, 300), (4096, 40)])
def test_key_owner_encryption_gives_the_public_path_bits(native, c_oracle, key_bits, batch):
    """phe_hip_encrypt_owner(_dev): r^n mod n^2 from r^n mod p^2 and r^n mod q^2 (half-exponentiation kernels with the
    exponent n), CRT lift, plaintext factor.  The same ciphertexts as phe_hip_encrypt and as the libgmp oracle."""
    rng = random.Random(key_bits)
    if key_bits == 4096:
        def prime(bits):
            while True:
                cand = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
                if pow(2, cand - 1, cand) == 1 and pow(3, cand - 1, cand) == 1:
                    return cand
        while True:
            p, q = prime(2048), prime(2048)
            if p != q and (p * q).bit_length() == 4096:
                break
        n_int, s1 = p * q, 128
        p, q, hp, hq, pinv = c_oracle.private_constants(n_int, p, q, s1, 64)
        ctx = native.Context(n_int, p, q, hp, hq, pinv, n_limbs=s1)
    else:
        g = load_golden(key_bits)
        n_int, s1 = H(g["n"]), key_bits // 32
        ctx = make_ctx(native, g)
        enc = g["raw_encrypt"]
        got = ctx.encrypt_owner(native.ints_to_limbs([H(e["m"]) % n_int for e in enc], s1),
                                native.ints_to_limbs([H(e["r"]) for e in enc], s1))
        assert native.limbs_to_ints(got) == [H(e["c"]) for e in enc]
    assert ctx.owner_encrypt_offered()
    m = native.ints_to_limbs([rng.randrange(0, n_int) for _ in range(batch)], s1)
    r = native.ints_to_limbs([rng.randrange(1, n_int) for _ in range(batch)], s1)
    c_owner = ctx.encrypt_owner(m, r)
    assert np.array_equal(c_owner, ctx.encrypt(m, r))
    idx = np.arange(0, batch, max(1, batch // 40))
    assert np.array_equal(c_owner[idx], c_oracle.encrypt(native.int_to_limbs(n_int, s1), m[idx], r[idx], nthreads=8))
    assert np.array_equal(ctx.decrypt(c_owner), m)
    pub_only = native.Context(n_int, n_limbs=s1)
    assert not pub_only.owner_encrypt_offered()
    with pytest.raises(ValueError):
        pub_only.encrypt_owner(m[:2], r[:2])


@pytest.mark.parametrize("key_bits,batch", [(2048, 40000), (3072, 9000), (4096, 1200)])
def test_scaled_modulus_path_gives_the_plain_path_bits(native, c_oracle, key_bits, batch, monkeypatch):
    """large batches take r^n modulo the scaled modulus n' = k*n (n' = -1 mod 2^29: quotient digits without a multiply)
    and one product pass back to n^2; PHE_HIP_NO_UNIT keeps the plain kernels: identical ciphertexts, and the oracle's"""
    rng = random.Random(key_bits + 1)
    if key_bits == 4096:
        def prime(bits):
            while True:
                cand = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
                if pow(2, cand - 1, cand) == 1 and pow(3, cand - 1, cand) == 1:
                    return cand
        while True:
            p, q = prime(2048), prime(2048)
            if p != q and (p * q).bit_length() == 4096:
                break
        n_int, s1 = p * q, 128
        make = lambda: native.Context(n_int, n_limbs=s1)
    else:
        g = load_golden(key_bits)
        n_int, s1 = H(g["n"]), key_bits // 32
        make = lambda: make_ctx(native, g, private=False)
    m = native.ints_to_limbs([rng.randrange(0, n_int) for _ in range(batch)], s1)
    r = native.ints_to_limbs([rng.randrange(1, n_int) for _ in range(batch)], s1)
    m[0], r[0] = 0, 0
    r[0, 0] = 1                                                   # r = 1
    r[1] = native.int_to_limbs(n_int - 1, s1)
    ctx = make()
    c = ctx.encrypt(m, r)
    idx = np.r_[0, 1, np.arange(2, batch, max(1, batch // 30))]
    n_arr = native.int_to_limbs(n_int, s1)
    assert np.array_equal(c[idx], c_oracle.encrypt(n_arr, m[idx], r[idx], nthreads=8))
    ob = ctx.obfuscate(c, r)
    assert np.array_equal(ob[idx], c_oracle.obfuscate(n_arr, c[idx], r[idx], nthreads=8))
    monkeypatch.setenv("PHE_HIP_NO_UNIT", "1")
    plain = make()
    assert np.array_equal(plain.encrypt(m, r), c)
    assert np.array_equal(plain.obfuscate(c, r), ob)
