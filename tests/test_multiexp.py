"""The encrypted dot product as one multi-exponentiation (include/phe_hip.h phe_hip_multiexp, csrc/split_core.h
multiexp_split_body, phe.EncryptedVector.dot).

What it must equal: the ciphertext the reference leaves after the chain  sum_i (c_i * k_i)  —
EncryptedNumber.__mul__ -> _raw_mul (phe/paillier.py:721-751, both branches), __add__ -> _add_encrypted (:677-703,
exponent alignment through decrease_exponent_to :570-601) -> _raw_add (:705-719) — bit for bit.

  * CPU (default run): the kernel body on the wave emulator against Python integers; the drop-in API on the
    emulator backend against its own scalar chain (which tests/test_api.py pins to the golden vectors) and against
    oracle/paillier_oracle.py's restatement of _raw_mul / _raw_add;
  * -m gpu: the C-ABI against the libgmp oracle, and at 2^16 elements through the plaintext identity
    D(prod c_i^k_i) = sum k_i m_i mod n.
"""
import functools
import random
import sys

import numpy as np
import pytest

from conftest import PKG, load_golden

if PKG not in sys.path:
    sys.path.insert(0, PKG)

from oracle.paillier_oracle import PyPublic, int_to_limbs, ints_to_limbs, limbs_to_ints  # noqa: E402


def H(x):
    return int(x, 16)


@pytest.fixture(scope="module")
def emu():
    from emu_lib import Emu
    return Emu()


@pytest.mark.parametrize("key_bits,group", [(256, 0), (1024, 0), (1024, 8), (2048, 0)])
def test_chunk_products_through_emulator(emu, key_bits, group):
    """k_multiexp_split's body: every chunk shares one ladder; ragged last chunk, exponent 0 and 1, bases 1 and
    n^2 - 1, all four window sizes; matrix form (several exponent rows on the tables of one task, row blocks that do
    not divide the row count) with per-entry selection of the inverted base"""
    emu.set_engine(True)
    emu.set_group(group)
    g = load_golden(key_bits)
    n_int = H(g["n"])
    nsq = n_int * n_int
    s1, s2 = key_bits // 32, key_bits // 16
    rng = random.Random(key_bits + group)
    n_arr = int_to_limbs(n_int, s1)
    shapes = [(7, 3, 64), (5, 1, 20), (9, 4, 130), (3, 8, 3)] if key_bits < 2048 else [(5, 2, 56)]
    for batch, chunk, ebits in shapes:
        bases = [rng.randrange(1, nsq) for _ in range(batch)]
        exps = [rng.getrandbits(ebits) for _ in range(batch)]
        exps[0], bases[1], bases[2] = 0, 1, nsq - 1
        if batch > 3:
            exps[3] = 1
        width = max(1, -(-ebits // 32))
        parts = emu.multiexp_n2(n_arr, ints_to_limbs(bases, s2), ints_to_limbs(exps, width), chunk)
        want = []
        for j in range(-(-batch // chunk)):
            v = 1
            for i in range(j * chunk, min(batch, (j + 1) * chunk)):
                v = v * pow(bases[i], exps[i], nsq) % nsq
            want.append(v)
        assert parts.shape == (len(want), 1, s2)
        assert limbs_to_ints(parts[:, 0]) == want, (batch, chunk, ebits)
    # matrix form
    batch, rows, chunk, row_block, ebits = (7, 5, 3, 2, 40) if key_bits < 2048 else (3, 2, 2, 2, 24)
    bases = [rng.randrange(1, nsq) for _ in range(batch)]
    invs = [pow(b, -1, nsq) for b in bases]
    ex = [[rng.getrandbits(ebits) for _ in range(batch)] for _ in range(rows)]
    ex[0][1] = 0
    neg = np.array([[rng.random() < 0.4 for _ in range(batch)] for _ in range(rows)], dtype=np.uint8)
    e_arr = np.stack([ints_to_limbs(row, 2) for row in ex])
    parts = emu.multiexp_n2(n_arr, ints_to_limbs(bases, s2), e_arr, chunk, base_inv=ints_to_limbs(invs, s2), neg=neg,
                            row_block=row_block)
    for j in range(-(-batch // chunk)):
        for r in range(rows):
            v = 1
            for i in range(j * chunk, min(batch, (j + 1) * chunk)):
                v = v * pow(invs[i] if neg[r][i] else bases[i], ex[r][i], nsq) % nsq
            assert limbs_to_ints(parts[j, r:r + 1]) == [v], (j, r)
    emu.set_group(0)


@pytest.mark.parametrize("key_bits", [256, 1024])
def test_chunk_products_on_pair_form_input(emu, key_bits):
    """multiexp_split_body with pair_in (round 4): the bases are resident rows in the pair form
    (to_pair_body's output) — no conversion in; same chunk products as on plain residues, one row and the matrix form"""
    emu.set_engine(True)
    emu.set_group(0)
    g = load_golden(key_bits)
    n_int = H(g["n"])
    nsq = n_int * n_int
    s1, s2 = key_bits // 32, key_bits // 16
    rng = random.Random(key_bits + 99)
    n_arr = int_to_limbs(n_int, s1)
    for batch, chunk, ebits in [(7, 3, 64), (5, 1, 20), (6, 4, 3)]:
        bases = [rng.randrange(1, nsq) for _ in range(batch)]
        exps = [rng.getrandbits(ebits) for _ in range(batch)]
        exps[0], bases[1], bases[2] = 0, 1, nsq - 1
        width = max(1, -(-ebits // 32))
        plain = emu.multiexp_n2(n_arr, ints_to_limbs(bases, s2), ints_to_limbs(exps, width), chunk)
        pairs = emu.pair_op(n_arr, 0, ints_to_limbs(bases, s2))
        assert pairs is not None and pairs.shape[1] != s2
        got = emu.multiexp_n2(n_arr, pairs, ints_to_limbs(exps, width), chunk, pair_in=True)
        assert np.array_equal(got, plain), (batch, chunk, ebits)
    batch, rows, chunk, row_block = 7, 5, 3, 2
    bases = [rng.randrange(1, nsq) for _ in range(batch)]
    ex = [[rng.getrandbits(40) for _ in range(batch)] for _ in range(rows)]
    e_arr = np.stack([ints_to_limbs(row, 2) for row in ex])
    plain = emu.multiexp_n2(n_arr, ints_to_limbs(bases, s2), e_arr, chunk, row_block=row_block)
    got = emu.multiexp_n2(n_arr, emu.pair_op(n_arr, 0, ints_to_limbs(bases, s2)), e_arr, chunk, row_block=row_block, pair_in=True)
    assert np.array_equal(got, plain)


def _chain(pub_oracle, cts, exps_c, encodings, exps_k):
    """the reference's left-to-right chain on integers: _raw_mul per term, then _add_encrypted's alignment + _raw_add"""
    nsq = pub_oracle.nsquare
    acc, acc_e = None, None
    for c, ec, k, ek in zip(cts, exps_c, encodings, exps_k):
        t, te = pub_oracle.raw_mul(c, k), ec + ek
        if acc is None:
            acc, acc_e = t, te
            continue
        if acc_e > te:                                       # phe/paillier.py:695-700
            acc, acc_e = pow(acc, 16 ** (acc_e - te), nsq), te
        elif te > acc_e:
            t = pow(t, 16 ** (te - acc_e), nsq)
        acc = pub_oracle.raw_add(acc, t)
    return acc, acc_e


@pytest.mark.parametrize("engine", [True, False])
def test_dot_equals_reference_chain(monkeypatch, emu, engine):
    import emu_backend
    emu_backend.install(monkeypatch)
    emu_backend.emu().set_engine(engine)
    from phe import paillier
    from phe.encoding import EncodedNumber
    g = load_golden(256)
    pub = paillier.PaillierPublicKey(H(g["n"]))
    priv = paillier.PaillierPrivateKey(pub, H(g["p"]), H(g["q"]))
    orc = PyPublic(pub.n)
    rs = [H(e["r"]) for e in g["raw_encrypt"][:7]]
    vals = np.array([0.5, -1.25, 3.0, 4.75, 1e-3, -7.0, 250.0])
    vec = pub.encrypt_batch(vals, r_values=rs)
    singles = [pub.encrypt(float(v), r_value=r) for v, r in zip(vals, rs)]
    cases = [
        np.array([2.0, -3.0, 0.5, 4.0, 1000.0, 0.25, -1e-2]),          # float64 array: mixed exponents and signs
        np.array([3, -4, 0, 1, 12345, -1, 7], dtype=np.int64),          # int64 array: exponent 0, both branches, 0 and 1
        [2, -3.5, 0.0, 1, 1e6, -0.125, 9],                              # list of Python numbers
        [EncodedNumber.encode(pub, 2.5), 3, -1, EncodedNumber.encode(pub, -0.75), 0.5, 1, 2],   # ready-made encodings
    ]
    for w in cases:
        got = vec.dot(w)
        terms = [s * (x if isinstance(x, EncodedNumber) else (x.item() if hasattr(x, "item") else x))
                 for s, x in zip(singles, w)]
        chain = functools.reduce(lambda a, b: a + b, terms)
        assert got.exponent == chain.exponent
        assert got.ciphertext(False) == chain.ciphertext(False)        # bit for bit the scalar chain
        encs = [x if isinstance(x, EncodedNumber) else EncodedNumber.encode(pub, x.item() if hasattr(x, "item") else x)
                for x in w]
        want, want_e = _chain(orc, [s.ciphertext(False) for s in singles], [s.exponent for s in singles],
                              [e.encoding for e in encs], [e.exponent for e in encs])
        assert (got.ciphertext(False), got.exponent) == (want, want_e)  # and the integer restatement of the reference
        expect = sum(float(v) * (x.decode() if isinstance(x, EncodedNumber) else float(x)) for v, x in zip(vals, w))
        assert abs(priv.decrypt(got) - expect) <= 1e-9 * max(1.0, abs(expect))
        assert got._EncryptedNumber__is_obfuscated is False
    assert vec.dot(2.0).ciphertext(False) == (vec * 2.0).sum().ciphertext(False)   # scalar broadcast
    assert vec.dot(cases[0]).ciphertext(False) == (vec * cases[0]).sum().ciphertext(False)
    with pytest.raises(ValueError):
        vec.dot([1.0, 2.0])
    with pytest.raises(NotImplementedError):
        vec.dot(singles)
    with pytest.raises(ValueError):
        vec[:0].dot([])
    emu_backend.emu().set_engine(True)


@pytest.mark.parametrize("engine", [True, False])
def test_matvec_rows_equal_dot(monkeypatch, emu, engine):
    """EncryptedVector.matvec: every row is bit for bit self.dot(row) (float rows with mixed exponents and signs, int
    rows, an all-positive matrix that needs no inverses); mean() = sum() / len"""
    import emu_backend
    emu_backend.install(monkeypatch)
    emu_backend.emu().set_engine(engine)
    from phe import paillier
    g = load_golden(256)
    pub = paillier.PaillierPublicKey(H(g["n"]))
    priv = paillier.PaillierPrivateKey(pub, H(g["p"]), H(g["q"]))
    vals = np.array([0.5, -1.25, 3.0, 4.75, 1e-3, -7.0, 250.0])
    vec = pub.encrypt_batch(vals, r_values=[H(e["r"]) for e in g["raw_encrypt"][:7]])
    rng = np.random.default_rng(1)
    for W in (rng.standard_normal((5, 7)), rng.integers(-50, 50, (4, 7)), rng.integers(0, 9, (3, 7)),
              np.array([[1e6, -1e-6, 0.0, 1.0, -1.0, 2.5, 0.125]])):
        out = vec.matvec(W)
        assert len(out) == W.shape[0]
        for r in range(W.shape[0]):
            d = vec.dot(W[r])
            assert (out[r].ciphertext(False), out[r].exponent) == (d.ciphertext(False), d.exponent), r
        got = priv.decrypt_batch(out)
        assert np.allclose(got, W.astype(np.float64) @ vals, rtol=1e-9, atol=1e-9)
    assert len(vec.matvec(np.zeros((0, 7)))) == 0
    W = rng.standard_normal((3, 7))
    assert (W @ vec).ciphertexts(False) == vec.matvec(W).ciphertexts(False)            # numpy defers to the vector
    assert (vec @ W.T).ciphertexts(False) == vec.matvec(W).ciphertexts(False)
    assert (W[0] @ vec).ciphertext(False) == (vec @ W[0]).ciphertext(False) == vec.dot(W[0]).ciphertext(False)
    with pytest.raises(ValueError):
        vec.matvec(np.zeros((2, 6)))
    assert abs(priv.decrypt(vec.mean()) - vals.mean()) < 1e-9
    emu_backend.emu().set_engine(True)


@pytest.mark.parametrize("key_bits", [256, 1024])
def test_table_lookup_form_through_emulator(emu, key_bits):
    """k_multiexp_tables + k_multiexp_lookup: CSR rows of different lengths (one of them empty) visited longest first,
    per-entry selection of the inverted base; dense rows without inverses"""
    emu.set_engine(True)
    g = load_golden(key_bits)
    n_int = H(g["n"])
    nsq = n_int * n_int
    s1, s2 = key_bits // 32, key_bits // 16
    rng = random.Random(key_bits)
    B, nnz = 7, [3, 0, 7, 1, 5, 2]
    bases = [rng.randrange(1, nsq) for _ in range(B)]
    invs = [pow(b, -1, nsq) for b in bases]
    cols, ex, ng = [], [], []
    for k in nnz:
        cs = rng.sample(range(B), k)
        cols += cs
        ex += [rng.getrandbits(rng.choice([1, 20, 56])) for _ in cs]
        ng += [rng.random() < 0.4 for _ in cs]
    row_ptr = np.cumsum([0] + nnz).astype(np.uint64)
    order = np.argsort(-np.array(nnz), kind="stable").astype(np.uint32)
    out = emu.multiexp_csr(int_to_limbs(n_int, s1), ints_to_limbs(bases, s2), ints_to_limbs(ex, 2),
                           base_inv=ints_to_limbs(invs, s2), neg=np.array(ng, dtype=np.uint8), row_ptr=row_ptr,
                           cols=np.array(cols, dtype=np.uint32), order=order)
    want, k = [], 0
    for cnt in nnz:
        v = 1
        for _ in range(cnt):
            v = v * pow(invs[cols[k]] if ng[k] else bases[cols[k]], ex[k], nsq) % nsq
            k += 1
        want.append(v)
    assert limbs_to_ints(out) == want
    dense = [[rng.getrandbits(40) for _ in range(B)] for _ in range(4)]
    out = emu.multiexp_csr(int_to_limbs(n_int, s1), ints_to_limbs(bases, s2),
                           np.concatenate([ints_to_limbs(r, 2) for r in dense]), rows=4)
    assert limbs_to_ints(out) == [functools.reduce(lambda a, b: a * b % nsq, [pow(b, e, nsq) for b, e in zip(bases, r)], 1)
                                  for r in dense]


def test_sparse_and_tall_matrices_take_the_table_form(monkeypatch, emu):
    """EncryptedVector.matvec with a scipy.sparse matrix (the stored entries of a row, as Bob.encrypted_score walks them)
    and with a dense matrix of many rows: every row bit for bit the dot product over its entries; an empty row is 1"""
    import scipy.sparse as sp
    import emu_backend
    emu_backend.install(monkeypatch)
    emu_backend.emu().set_engine(True)
    from phe import paillier
    g = load_golden(256)
    pub = paillier.PaillierPublicKey(H(g["n"]))
    priv = paillier.PaillierPrivateKey(pub, H(g["p"]), H(g["q"]))
    vals = np.array([0.5, -1.25, 3.0, 4.75, 1e-3, -7.0, 250.0])
    vec = pub.encrypt_batch(vals, r_values=[H(e["r"]) for e in g["raw_encrypt"][:7]])
    rng = np.random.default_rng(2)
    D = rng.standard_normal((6, 7))
    D[rng.random((6, 7)) < 0.6] = 0.0
    D[1, :] = 0.0
    for M in (sp.csr_matrix(D), sp.coo_matrix(D), sp.csr_matrix(np.rint(D * 10).astype(np.int64))):
        dense = M.toarray()
        out = vec.matvec(M)
        for r in range(6):
            nz = np.nonzero(dense[r])[0]
            if len(nz) == 0:
                assert out[r].ciphertext(False) == 1 and priv.decrypt(out[r]) == 0
                continue
            sub = paillier.EncryptedVector.from_numbers(pub, [vec[int(i)] for i in nz])
            d = sub.dot(dense[r][nz])
            assert (out[r].ciphertext(False), out[r].exponent) == (d.ciphertext(False), d.exponent), r
        assert np.allclose(priv.decrypt_batch(out), dense.astype(np.float64) @ vals, rtol=1e-9, atol=1e-12)
    from phe import ciphertext
    monkeypatch.setattr(ciphertext, "TABLE_COMPACT_BYTES", 0)         # host vectors: tables only for the stored columns
    E = D.copy()
    E[:, [0, 3, 5]] = 0.0
    compact = vec.matvec(sp.csr_matrix(E))
    for r in range(6):
        nz = np.nonzero(E[r])[0]
        if len(nz):
            sub = paillier.EncryptedVector.from_numbers(pub, [vec[int(i)] for i in nz])
            assert compact[r].ciphertext(False) == sub.dot(E[r][nz]).ciphertext(False)
    assert np.allclose(priv.decrypt_batch(compact), E @ vals, rtol=1e-9, atol=1e-12)
    W = rng.integers(-20, 20, (70, 7))                       # >= 64 rows: dense rows on the shared tables
    out = vec.matvec(W)
    for r in (0, 33, 69):
        d = vec.dot(W[r])
        assert (out[r].ciphertext(False), out[r].exponent) == (d.ciphertext(False), d.exponent)
    assert np.allclose(priv.decrypt_batch(out), W.astype(np.float64) @ vals, rtol=1e-9)


# ---- GPU -------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def native():
    from phe import _native
    assert _native.device_count() >= 1
    return _native


def _ctx(native, g, private=False):
    if private:
        return native.Context(H(g["n"]), H(g["p"]), H(g["q"]), H(g["hp"]), H(g["hq"]), H(g["p_inverse"]),
                              n_limbs=g["key_bits"] // 32)
    return native.Context(H(g["n"]), n_limbs=g["key_bits"] // 32)


@pytest.mark.gpu
@pytest.mark.parametrize("engine", ["split", "full"])
@pytest.mark.parametrize("key_bits", [1024, 2048, 3072])
def test_multiexp_vs_gmp_oracle(native, c_oracle, monkeypatch, key_bits, engine):
    """phe_hip_multiexp against the per-element powmod of the libgmp oracle folded with Python integers: batches of
    1 .. 700, exponents of 1 .. 200 bits (all window sizes), forced chunk sizes (ragged last chunk) and the automatic one"""
    monkeypatch.setenv("PHE_HIP_ENGINE", engine)
    g = load_golden(key_bits)
    n_int = H(g["n"])
    nsq = n_int * n_int
    s1, s2 = key_bits // 32, key_bits // 16
    n = native.int_to_limbs(n_int, s1)
    ctx = _ctx(native, g)
    rng = random.Random(key_bits)
    one = np.zeros((1, s2), np.uint32)
    one[0, 0] = 1
    assert np.array_equal(ctx.multiexp(np.zeros((0, s2), np.uint32), np.zeros((0, 1), np.uint32)), one)
    shapes = [(1, 64, "0"), (5, 3, "0"), (37, 56, "5"), (300, 64, "0"), (300, 130, "7"), (700, 20, "16"), (64, 200, "3")]
    if key_bits == 3072:
        shapes = shapes[:4]
    for batch, ebits, chunk in shapes:
        monkeypatch.setenv("PHE_HIP_MULTI_CHUNK", chunk)
        bases = native.ints_to_limbs([rng.randrange(1, nsq) for _ in range(batch)], s2)
        exps = [rng.getrandbits(ebits) for _ in range(batch)]
        exps[0] = 0
        e = native.ints_to_limbs(exps, max(1, -(-ebits // 32)))
        e_full = native.ints_to_limbs(exps, s1)
        terms = native.limbs_to_ints(c_oracle.mul(n, bases, e_full, nthreads=8))
        want = functools.reduce(lambda a, b: a * b % nsq, terms, 1)
        assert native.limbs_to_ints(ctx.multiexp(bases, e)) == [want], (batch, ebits, chunk)


@pytest.mark.gpu
def test_dot_large_batch_properties(native, c_oracle):
    """2048-bit key, 2^16 ciphertexts (resident), int64 and float64 weights: the dot product decrypts to
    sum k_i m_i, equals the `*` / sum() composition bit for bit, and a 512-element prefix equals the oracle chain"""
    import phe
    from phe import paillier
    g = load_golden(2048)
    pub = paillier.PaillierPublicKey(H(g["n"]))
    priv = paillier.PaillierPrivateKey(pub, H(g["p"]), H(g["q"]))
    B = 1 << 16
    rs = np.random.Generator(np.random.PCG64(99))
    vals = rs.integers(-10 ** 6, 10 ** 6, B)
    vec = pub.encrypt_batch(vals, device=True)
    w_int = rs.integers(-2 ** 40, 2 ** 40, B)
    got = vec.dot(w_int)
    assert priv.decrypt(got) == int(np.dot(vals.astype(object), w_int.astype(object)))
    assert got.ciphertext(False) == (vec * w_int).sum().ciphertext(False)
    w_f = rs.standard_normal(B)
    got_f = vec.dot(w_f)
    assert got_f.ciphertext(False) == (vec * w_f).sum().ciphertext(False)
    expect = float(np.dot(vals.astype(np.float64), w_f))
    assert abs(priv.decrypt(got_f) - expect) <= 1e-6 * max(1.0, abs(expect))
    # a prefix against the libgmp oracle
    k = 512
    n_int = pub.n
    nsq = n_int * n_int
    head = vec[:k].to_host()
    c = head._limbs
    mag = np.abs(w_int[:k]).astype(np.uint64)
    neg = w_int[:k] < 0
    n = native.int_to_limbs(n_int, 64)
    scal = native.ints_to_limbs([n_int - int(m) if s else int(m) for m, s in zip(mag, neg)], 64)
    terms = native.limbs_to_ints(c_oracle.mul(n, c, scal, nthreads=8))        # _raw_mul incl. the inverse branch
    want = functools.reduce(lambda a, b: a * b % nsq, terms, 1)
    assert head.dot(w_int[:k]).ciphertext(False) == want
    assert isinstance(got, phe.EncryptedNumber)


@pytest.mark.gpu
def test_dot_and_matvec_on_vectors_in_the_pair_form(native, c_oracle):
    """`vec.to_pair().dot(w)` / `.matvec(W)` with non-negative weights read the resident pair rows as they are
    (phe_hip_pair_multiexp_rows_dev): the ciphertext bits of the plain-residue path, the vector left in its form; a negative
    weight falls back to the residues (the inverse branch of _raw_mul), same bits again; forced chunk sizes as well"""
    from phe import paillier
    g = load_golden(2048)
    pub = paillier.PaillierPublicKey(H(g["n"]))
    priv = paillier.PaillierPrivateKey(pub, H(g["p"]), H(g["q"]))
    rs = np.random.Generator(np.random.PCG64(41))
    B = 5000
    vals = rs.integers(-10 ** 6, 10 ** 6, B)
    vec = pub.encrypt_batch(vals, device=True)
    pv = vec.to_pair()
    assert pv._pair and pv._store.cols != vec._store.cols
    w = rs.integers(0, 2 ** 56, B)                                   # non-negative: stays in the form
    got = pv.dot(w)
    assert pv._pair                                                  # not converted back
    assert got.ciphertext(False) == vec.dot(w).ciphertext(False)
    assert priv.decrypt(got) == int(np.dot(vals.astype(object), w.astype(object)))
    wf = np.abs(rs.standard_normal(B))
    assert pv.dot(wf).ciphertext(False) == vec.dot(wf).ciphertext(False)
    w_neg = w.copy()
    w_neg[7] = -5
    assert pv.dot(w_neg).ciphertext(False) == vec.dot(w_neg).ciphertext(False)
    W = rs.integers(0, 2 ** 40, (6, B))
    rows_pair = pv.matvec(W)
    rows_plain = vec.matvec(W)
    assert rows_pair.ciphertexts(False) == rows_plain.ciphertexts(False)
    assert priv.decrypt_batch(rows_pair) == [int(np.dot(vals.astype(object), r.astype(object))) for r in W]


@pytest.mark.gpu
@pytest.mark.parametrize("forced", [None, ("5", "3"), ("1", "64"), ("16", "1")])
def test_matvec_against_row_dots_and_plaintext(monkeypatch, forced):
    """the matrix form through the C-ABI, host and device-resident vectors, automatic and forced (chunk, row block)
    shapes incl. blocks that do not divide the row count: every row equals dot(row) bit for bit and decrypts to W @ x"""
    from phe import paillier
    if forced:
        monkeypatch.setenv("PHE_HIP_MULTI_CHUNK", forced[0])
        monkeypatch.setenv("PHE_HIP_MULTI_ROWBLOCK", forced[1])
    g = load_golden(1024)
    pub = paillier.PaillierPublicKey(H(g["n"]))
    priv = paillier.PaillierPrivateKey(pub, H(g["p"]), H(g["q"]))
    rs = np.random.Generator(np.random.PCG64(17))
    x = rs.standard_normal(93)
    host = pub.encrypt_batch(x)
    dev = host.to_device()
    for W in (rs.standard_normal((41, 93)), rs.integers(-1000, 1000, (7, 93)), rs.integers(0, 1 << 40, (3, 93))):
        a, b = host.matvec(W), dev.matvec(W)
        assert b.on_device and a.ciphertexts(False) == b.ciphertexts(False) and a.exponents == b.exponents
        for r in (0, W.shape[0] // 2, W.shape[0] - 1):
            d = host.dot(W[r])
            assert (a[r].ciphertext(False), a[r].exponent) == (d.ciphertext(False), d.exponent)
        want = W.astype(np.float64) @ x
        assert np.allclose(priv.decrypt_batch(b), want, rtol=1e-9, atol=1e-6)


@pytest.mark.gpu
def test_matvec_without_split_engine_goes_row_by_row(monkeypatch):
    """PHE_HIP_ENGINE=full (or a key width without a split geometry): the C entry refuses the matrix form and
    Engine.raw_matvec calls the single-row entry point per row — same bits as with the split engine"""
    from phe import paillier
    g = load_golden(1024)
    rs = np.random.Generator(np.random.PCG64(5))
    x, W = rs.standard_normal(21), rs.standard_normal((4, 21))
    r_values = [int(v) for v in rs.integers(2, 1 << 62, 21)]
    ref = paillier.PaillierPublicKey(H(g["n"])).encrypt_batch(x, r_values=r_values).matvec(W)
    monkeypatch.setenv("PHE_HIP_ENGINE", "full")
    pub = paillier.PaillierPublicKey(H(g["n"]))
    assert pub._get_engine().ctx.info()["engine_pub"] != "split"
    for vec in (pub.encrypt_batch(x, r_values=r_values), pub.encrypt_batch(x, r_values=r_values, device=True)):
        out = vec.matvec(W)
        assert out.ciphertexts(False) == ref.ciphertexts(False) and out.exponents == ref.exponents


@pytest.mark.gpu
def test_sparse_matvec_on_gpu():
    """scipy.sparse rows (5 % dense, ragged, some empty) and a tall dense matrix through phe_hip_multiexp_csr_dev, host and
    device-resident vectors: equal to the chunked matrix form row for row, and to W @ x after decryption"""
    import scipy.sparse as sp
    from phe import paillier
    g = load_golden(1024)
    pub = paillier.PaillierPublicKey(H(g["n"]))
    priv = paillier.PaillierPrivateKey(pub, H(g["p"]), H(g["q"]))
    rs = np.random.Generator(np.random.PCG64(23))
    x = rs.standard_normal(400)
    host = pub.encrypt_batch(x)
    dev = host.to_device()
    D = rs.standard_normal((300, 400))
    D[rs.random((300, 400)) < 0.95] = 0.0
    D[7, :] = 0.0
    S = sp.csr_matrix(D)
    a, b = host.matvec(S), dev.matvec(S)
    assert b.on_device and a.ciphertexts(False) == b.ciphertexts(False) and a.exponents == b.exponents
    assert a[7].ciphertext(False) == 1
    assert np.allclose(priv.decrypt_batch(b), D @ x, rtol=1e-9, atol=1e-9)
    for r in (0, 150, 299):
        nz = np.nonzero(D[r])[0]
        sub = paillier.EncryptedVector.from_numbers(pub, [host[int(i)] for i in nz])
        d = sub.dot(D[r][nz])
        assert (a[r].ciphertext(False), a[r].exponent) == (d.ciphertext(False), d.exponent)
    W = rs.integers(-1000, 1000, (130, 400))                 # tall and dense: the table form; 5 rows: the chunked form
    tall = dev.matvec(W)
    few = dev.matvec(W[:5])
    assert tall[:5].ciphertexts(False) == few.ciphertexts(False)
    assert np.allclose(priv.decrypt_batch(tall), W.astype(np.float64) @ x, rtol=1e-9, atol=1e-6)
