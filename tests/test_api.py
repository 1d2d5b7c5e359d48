"""The drop-in `phe` API (python-paillier_amd/phe), exercised through both backends:

  * backend "emu" (CPU, default run): phe._native.Context replaced by tests/emu_backend.EmuContext, i.e. the
    device algorithm on the wave emulator — checks the host-side logic (encoding, exponent alignment,
    obfuscation state machine, error types, batched containers) on a GPU-less box;
  * backend "hip" (-m gpu): the real C-ABI / HIP kernels.

Expected values come from tests/golden (real reference) and from the behaviours the reference's own
tests pin (cited inline as phe/tests/paillier_test.py:<line>).
"""
import math
import pickle
import sys

import numpy as np
import pytest

from conftest import PKG, load_golden, load_kat

if PKG not in sys.path:
    sys.path.insert(0, PKG)

import phe  # noqa: E402
from phe import paillier  # noqa: E402


def H(x):
    return int(x, 16)


@pytest.fixture(params=["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request, monkeypatch):
    if request.param == "emu":
        import emu_backend
        emu_backend.install(monkeypatch)
    return request.param


@pytest.fixture
def keys(backend):
    g = load_golden(256)
    pub = paillier.PaillierPublicKey(H(g["n"]))
    priv = paillier.PaillierPrivateKey(pub, H(g["q"]), H(g["p"]))     # unordered on purpose
    return g, pub, priv


def test_public_names():
    for name in ["EncodedNumber", "generate_paillier_keypair", "EncryptedNumber", "PaillierPrivateKey",
                 "PaillierPublicKey", "PaillierPrivateKeyring", "util", "paillier", "encoding"]:
        assert hasattr(phe, name)
    assert paillier.DEFAULT_KEYSIZE == 3072


def test_private_key_constants_match_reference():
    g = load_golden(1024)
    pub = paillier.PaillierPublicKey(H(g["n"]))
    priv = paillier.PaillierPrivateKey(pub, H(g["q"]), H(g["p"]))
    assert (priv.p, priv.q, priv.hp, priv.hq, priv.p_inverse) == tuple(H(g[k]) for k in ("p", "q", "hp", "hq", "p_inverse"))
    assert priv == paillier.PaillierPrivateKey(pub, H(g["p"]), H(g["q"]))       # paillier_test.py:82-86
    with pytest.raises(ValueError):
        paillier.PaillierPrivateKey(pub, H(g["p"]), H(g["q"]) + 2)
    totient = (priv.p - 1) * (priv.q - 1)
    assert paillier.PaillierPrivateKey.from_totient(pub, totient) == priv        # paillier_test.py:73-80
    assert pickle.loads(pickle.dumps(priv)) == priv


def test_encoding_matches_reference_fixture():
    """encode() exponents / encodings against ciphertexts the real reference produced (host logic only)."""
    for bits in (256, 1024):
        g = load_golden(bits)
        pub = paillier.PaillierPublicKey(H(g["n"]))
        from oracle.paillier_oracle import PyPublic, PyPrivate
        opub = PyPublic(pub.n)
        opriv = PyPrivate(opub, H(g["p"]), H(g["q"]))
        for e in g["encrypt_api"]:
            value = eval(e["value"])
            enc = phe.EncodedNumber.encode(pub, value)
            assert enc.exponent == e["exponent"]
            assert opub.raw_encrypt(enc.encoding, H(e["r"])) == H(e["c"])
            back = phe.EncodedNumber(pub, opriv.raw_decrypt(H(e["c"])), e["exponent"]).decode()
            assert repr(back) == e["decrypted"]


def test_encode_many_equals_scalar_encode():
    pub = paillier.PaillierPublicKey(H(load_golden(1024)["n"]))
    rs = np.random.Generator(np.random.PCG64(3))
    floats = np.concatenate([rs.standard_normal(200) * 10.0 ** rs.integers(-30, 30, 200), [0.0, -0.0, 5e-324, 1.5]])
    encs, exps = phe.EncodedNumber.encode_many(pub, floats)
    for v, a, b in zip(floats.tolist(), encs, exps):
        s = phe.EncodedNumber.encode(pub, v)
        assert (s.encoding, s.exponent) == (a, b)
    ints = rs.integers(-2 ** 62, 2 ** 62, 100)
    encs, exps = phe.EncodedNumber.encode_many(pub, ints)
    assert exps == [0] * 100 and encs == [int(v) % pub.n for v in ints]
    assert phe.EncodedNumber.decode_many(pub, encs, exps) == ints.tolist()
    with pytest.raises(ValueError):
        phe.EncodedNumber.encode_many(paillier.PaillierPublicKey(126869), np.array([1e9]))


def test_encoded_number_errors():
    pub = paillier.PaillierPublicKey(H(load_golden(256)["n"]))
    with pytest.raises(TypeError):
        phe.EncodedNumber.encode(pub, "1")
    with pytest.raises(TypeError):
        phe.EncodedNumber.encode(pub, np.int64(3))            # np.int64 is not int
    with pytest.raises(ValueError):
        phe.EncodedNumber.encode(pub, pub.max_int + 1)
    with pytest.raises(OverflowError):
        phe.EncodedNumber(pub, pub.max_int + 5, 0).decode()
    with pytest.raises(ValueError):
        phe.EncodedNumber(pub, pub.n, 0).decode()
    e = phe.EncodedNumber.encode(pub, 1.5)
    with pytest.raises(ValueError):
        e.decrease_exponent_to(e.exponent + 1)
    assert e.decrease_exponent_to(e.exponent - 3).decode() == 1.5


def test_raw_kat(backend):
    k = load_kat()
    pub = paillier.PaillierPublicKey(k["n"])
    priv = paillier.PaillierPrivateKey(pub, k["p"], k["q"])
    assert pub.raw_encrypt(k["m"], k["r"]) == k["c"]                              # paillier_test.py:128-136
    assert priv.raw_decrypt(k["c"]) == k["m"]
    assert pub.encrypt(k["m"], r_value=k["r"]).ciphertext(False) == k["c"]        # :138-142
    assert pub.encrypt(1, r_value=1).ciphertext(False) == k["encrypt_1_r_1"]      # :144-149
    assert pub.encrypt(1).ciphertext(False) != k["encrypt_1_r_1"]                 # random r
    with pytest.raises(TypeError):
        pub.raw_encrypt("123")                                                    # :157-164
    with pytest.raises(TypeError):
        priv.raw_decrypt("935906717")


def test_modulo_n_wrap(keys):
    _, pub, priv = keys
    for m, want in ((pub.n - 1, pub.n - 1), (pub.n, 0), (pub.n + 1, 1)):          # paillier_test.py:114-126
        assert priv.raw_decrypt(pub.raw_encrypt(m)) == want


def test_golden_api_vectors(keys):
    g, pub, priv = keys
    for e in g["encrypt_api"]:
        value = eval(e["value"])
        enc = pub.encrypt(value, r_value=H(e["r"]))
        assert enc.ciphertext(False) == H(e["c"]) and enc.exponent == e["exponent"]
        assert repr(priv.decrypt(enc)) == e["decrypted"]


def test_obfuscation_state_machine(keys):
    _, pub, priv = keys
    a = pub.encrypt(3.25, r_value=12345)                   # explicit r: not flagged (paillier_test.py:1023-1039)
    assert a._EncryptedNumber__is_obfuscated is False
    c0 = a.ciphertext(be_secure=False)
    c1 = a.ciphertext()                                    # obfuscates in place, once
    assert a._EncryptedNumber__is_obfuscated is True and c1 != c0 and a.ciphertext() == c1
    assert priv.decrypt(a) == 3.25
    b = pub.encrypt(3.25)
    assert b._EncryptedNumber__is_obfuscated is True
    s = a + b
    assert s._EncryptedNumber__is_obfuscated is False      # results of + and * are not obfuscated
    assert (a * 2)._EncryptedNumber__is_obfuscated is False
    z = pub.encrypt(7, r_value=0)
    assert z.ciphertext(False) == (1 + pub.n * 7) % pub.nsquare and z._EncryptedNumber__is_obfuscated is False


def test_homomorphic_arithmetic(keys):
    _, pub, priv = keys
    a, b = pub.encrypt(15), pub.encrypt(-1.125)
    assert priv.decrypt(a + b) == 13.875
    assert priv.decrypt(a + 5) == 20 and priv.decrypt(5 + a) == 20 and priv.decrypt(a + 0.5) == 15.5
    assert priv.decrypt(a - b) == 16.125 and priv.decrypt(1 - a) == -14 and priv.decrypt(a - 20) == -5
    assert priv.decrypt(a * 3) == 45 and priv.decrypt(-2 * a) == -30 and priv.decrypt(b * 0.5) == -0.5625
    assert priv.decrypt(a / 4) == 3.75
    assert priv.decrypt(a * 0) == 0
    c = a * 1
    assert c.ciphertext(False) == a.ciphertext(False)                                 # paillier_test.py:893-899
    assert priv.decrypt(a + phe.EncodedNumber.encode(pub, 2.5)) == 17.5
    assert priv.decrypt(a * phe.EncodedNumber.encode(pub, -2)) == -30
    assert priv.decrypt(sum([a, b, a])) == 28.875                                     # __radd__ with 0
    with pytest.raises(NotImplementedError):
        a * b
    before = (a.ciphertext(False), a.exponent)
    _ = a + pub.encrypt(0.001)
    assert (a.ciphertext(False), a.exponent) == before                                # operands untouched (:780-790)


def test_exponent_alignment_and_decrease(keys):
    _, pub, priv = keys
    a = pub.encrypt(1.0 / 3)
    b = pub.encrypt(2 ** 40)
    s = a + b
    assert s.exponent == min(a.exponent, b.exponent)
    assert math.isclose(priv.decrypt(s), 2 ** 40 + 1.0 / 3, rel_tol=1e-15)
    low = a.decrease_exponent_to(a.exponent - 5)
    assert low.exponent == a.exponent - 5 and priv.decrypt(low) == priv.decrypt(a)
    with pytest.raises(ValueError):
        a.decrease_exponent_to(a.exponent + 1)


def test_key_mismatch_and_type_errors(keys):
    g, pub, priv = keys
    g2 = load_golden(1024)
    pub2 = paillier.PaillierPublicKey(H(g2["n"]))
    priv2 = paillier.PaillierPrivateKey(pub2, H(g2["p"]), H(g2["q"]))
    a = pub.encrypt(1, r_value=1)
    with pytest.raises(ValueError):
        priv2.decrypt(a)                                                              # paillier_test.py:468-473
    with pytest.raises(TypeError):
        priv.decrypt(a.ciphertext(False))
    a2 = paillier.EncryptedNumber(pub2, 5, 0)
    with pytest.raises(ValueError):
        a + a2
    with pytest.raises(ValueError):
        a + phe.EncodedNumber.encode(pub2, 1)
    with pytest.raises(ValueError):
        a._raw_mul(pub.n)
    with pytest.raises(ValueError):
        a._raw_mul(-1)
    with pytest.raises(TypeError):
        a._raw_mul(1.5)
    with pytest.raises(TypeError):
        paillier.EncryptedNumber("key", 5)


def test_overflow_detection(keys):
    _, pub, priv = keys
    big = pub.encrypt(pub.max_int)
    with pytest.raises(OverflowError):
        priv.decrypt(big + big)                                                       # paillier_test.py:608-635


def test_keyring(keys):
    _, pub, priv = keys
    ring = paillier.PaillierPrivateKeyring()
    ring.add(priv)
    assert len(ring) == 1 and ring[pub] == priv
    assert ring.decrypt(pub.encrypt(25, r_value=3)) == 25
    with pytest.raises(TypeError):
        ring.add(pub)
    del ring[pub]
    with pytest.raises(KeyError):
        ring.decrypt(pub.encrypt(1, r_value=3))


def test_numpy_interop(keys):
    _, pub, priv = keys
    vals = [0.5, -1.25, 3.0, 4.75]
    enc = [pub.encrypt(v) for v in vals]
    assert priv.decrypt(np.mean(enc)) == np.mean(vals)                                # math_test.py:26-42
    w = [2.0, -1.0, 0.5, 4]
    assert priv.decrypt(np.dot(enc, w)) == float(np.dot(vals, w))                     # math_test.py:44-58


def test_encrypted_vector(keys):
    g, pub, priv = keys
    vals = np.array([0.5, -1.25, 3.0, 4.75, 1e-3, -7.0])
    rs = [H(e["r"]) for e in g["raw_encrypt"][3:9]]
    vec = pub.encrypt_batch(vals, r_values=rs)
    singles = [pub.encrypt(float(v), r_value=r) for v, r in zip(vals, rs)]
    assert vec.ciphertexts(be_secure=False) == [s.ciphertext(False) for s in singles]   # batch == scalar path, bit for bit
    assert vec.exponents == [s.exponent for s in singles]
    assert priv.decrypt_batch(vec) == vals.tolist()
    assert priv.decrypt(vec[2]) == 3.0 and len(vec[1:3]) == 2
    w = [2.0, -3.0, 0.5, 4, 1000.0, 0.25]
    prod = vec * w
    want = [priv.decrypt(s * x) for s, x in zip(singles, w)]
    assert prod.ciphertexts(False) == [(s * x).ciphertext(False) for s, x in zip(singles, w)]
    assert priv.decrypt_batch(prod) == want
    other = pub.encrypt_batch(np.array([1.0, 2.0, 3.0, 4.0, 5.0, 6.0]), r_values=rs)
    total = vec + other
    assert total.ciphertexts(False) == [(a + b).ciphertext(False) for a, b in zip(singles, other.to_numbers())]
    assert priv.decrypt_batch(vec + 1.5) == [v + 1.5 for v in vals.tolist()]
    assert priv.decrypt_batch(vec - other) == [a - b for a, b in zip(vals.tolist(), [1.0, 2.0, 3.0, 4.0, 5.0, 6.0])]
    assert math.isclose(priv.decrypt(vec.sum()), float(vals.sum()), rel_tol=1e-12)
    assert math.isclose(priv.decrypt(vec.dot(w)), float(np.dot(vals, w)), rel_tol=1e-12)
    fresh = pub.encrypt_batch([1, 2, 3])
    assert all(fresh._obfuscated) and priv.decrypt_batch(fresh) == [1, 2, 3]
    un = pub.encrypt_batch([1, 2, 3], r_values=[1, 1, 1])
    before = un.ciphertexts(be_secure=False)
    after = un.ciphertexts()                        # be_secure: obfuscates the whole vector once
    assert before != after and all(un._obfuscated) and un.ciphertexts() == after
    assert priv.decrypt_batch(un) == [1, 2, 3]
    with pytest.raises(ValueError):
        priv.decrypt_batch(paillier.EncryptedVector(paillier.PaillierPublicKey(H(load_golden(1024)["n"])),
                                                    np.zeros((1, 64), np.uint32), [0]))
    # the reference's documented JSON vector format (docs/serialisation.rst:24-43)
    import json
    text = vec.to_json(be_secure=False)
    doc = json.loads(text)
    assert int(doc["public_key"]["n"]) == pub.n
    assert [(int(c), e) for c, e in doc["values"]] == list(zip(vec.ciphertexts(False), vec.exponents))
    back = paillier.EncryptedVector.from_json(text)
    assert back.public_key == pub and priv.decrypt_batch(back) == vals.tolist()
    singles_rec = [paillier.EncryptedNumber(back.public_key, int(c), int(e)) for c, e in doc["values"]]   # the doc's recipe
    assert [priv.decrypt(x) for x in singles_rec] == vals.tolist()


def test_plaintext_array_with_much_larger_exponents_adds_like_the_scalar_path(backend):
    """vec + ndarray where the plaintexts' natural exponents sit far above the ciphertexts': the shifted mantissas exceed
    64 bits and are formed as limb rows (codec.signed_limbs_to_plain) — bit for bit EncryptedNumber.__add__ per element"""
    import random
    from phe import _native
    from phe.codec import EncodedNumber
    g = load_golden(1024)
    pub = paillier.PaillierPublicKey(H(g["n"]))
    priv = paillier.PaillierPrivateKey(pub, H(g["p"]), H(g["q"]))
    rng = np.random.default_rng(3)
    x = rng.random(12) * 1e-6
    w = rng.standard_normal(12) * 1e3
    w[0], w[1] = 0.0, -1e3
    vec = pub.encrypt_batch(x, r_values=[int(v) for v in rng.integers(2, 1 << 60, 12)])
    out = vec + w
    for i, single in enumerate(vec.to_numbers()):
        ref = single + float(w[i])
        assert (out[i].ciphertext(False), out[i].exponent) == (ref.ciphertext(False), ref.exponent), i
    assert np.allclose(priv.decrypt_batch(out), x + w)
    r = random.Random(1)
    vals = [r.getrandbits(r.choice([1, 40, 64, 65, 100, 130])) for _ in range(200)] + [0, 1, (1 << 64) - 1, 1 << 64, 1 << 96]
    neg = np.array([r.random() < 0.5 for _ in vals])
    got = EncodedNumber.signed_limbs_to_plain(pub, _native.ints_to_limbs(vals, 5), neg, 32)
    assert _native.limbs_to_ints(got) == [(pub.n - v) if (s and v) else v for v, s in zip(vals, neg)]


def test_concatenate_vectors(keys):
    g, pub, priv = keys
    rs = [H(e["r"]) for e in g["raw_encrypt"][:6]]
    a = pub.encrypt_batch(np.array([0.5, -1.25, 3.0]), r_values=rs[:3])
    b = pub.encrypt_batch([7, -8], r_values=rs[3:5])
    c = pub.encrypt_batch([1.5])                                         # fresh: obfuscated
    whole = paillier.EncryptedVector.concatenate([a, b, a[:0], c])
    assert len(whole) == 6 and whole.ciphertexts(False)[:5] == a.ciphertexts(False) + b.ciphertexts(False)
    assert whole.exponents == a.exponents + b.exponents + c.exponents
    assert whole._obfuscated.tolist() == [False] * 5 + [True]
    assert priv.decrypt_batch(whole) == [0.5, -1.25, 3.0, 7, -8, 1.5]
    picked = whole[[4, 0, 0]]
    assert picked.ciphertexts(False) == [whole.ciphertexts(False)[k] for k in (4, 0, 0)] and picked.exponents == [0, a.exponents[0]] + [a.exponents[0]]
    assert priv.decrypt_batch(whole[np.array([True, False, True, False, False, True])]) == [0.5, 3.0, 1.5]
    other = paillier.PaillierPublicKey(H(load_golden(1024)["n"]))
    with pytest.raises(ValueError):
        paillier.EncryptedVector.concatenate([a, other.encrypt_batch([1], r_values=[1])])
    with pytest.raises(ValueError):
        paillier.EncryptedVector.concatenate([])


def test_array_operands_take_the_same_path_as_lists(keys):
    """numpy operands are encoded without a Python integer per element (codec array forms); the ciphertext bits must
    be the ones the element-by-element path produces."""
    g, pub, priv = keys
    rng = np.random.Generator(np.random.PCG64(3))
    vals = rng.standard_normal(9) * 10.0 ** rng.integers(-4, 5, 9)
    rs = [H(e["r"]) for e in g["raw_encrypt"][3:12]]
    vec = pub.encrypt_batch(vals, r_values=rs)
    assert vec.ciphertexts(False) == pub.encrypt_batch(vals.tolist(), r_values=rs).ciphertexts(False)
    w = rng.standard_normal(9)
    w[0], w[1], w[2] = 0.0, 1.0, -1.0
    assert (vec * w).ciphertexts(False) == (vec * w.tolist()).ciphertexts(False)
    assert (vec * w).exponents == (vec * w.tolist()).exponents
    k = rng.integers(-1000, 1000, 9)
    assert (vec * k).ciphertexts(False) == (vec * [int(x) for x in k]).ciphertexts(False)
    b = rng.standard_normal(9) * 10.0 ** rng.integers(-6, 3, 9)       # exponents above and below the vector's
    assert (vec + b).ciphertexts(False) == (vec + b.tolist()).ciphertexts(False)
    assert (vec + b).exponents == (vec + b.tolist()).exponents
    assert (vec - b).ciphertexts(False) == (vec - b.tolist()).ciphertexts(False)
    assert (vec + k).ciphertexts(False) == (vec + [int(x) for x in k]).ciphertexts(False)
    got = priv.decrypt_batch(vec * w + b)
    assert all(math.isclose(a, v * x + y, rel_tol=1e-9, abs_tol=1e-12) for a, v, x, y in zip(got, vals, w, b))
    ints = pub.encrypt_batch(np.array([5, -6, 7], dtype=np.int64), r_values=rs[:3])
    assert priv.decrypt_batch(ints) == [5, -6, 7] and ints.exponents == [0, 0, 0]
    assert priv.decrypt_batch(ints * np.array([2, -3, 0])) == [10, 18, 0]


@pytest.mark.gpu
def test_device_resident_vector_matches_host_vector():
    """EncryptedVector(device=True): every op stays in HBM and gives the same ciphertext bits as the host-array path
    (which in turn equals the scalar path, test_encrypted_vector)."""
    g = load_golden(1024)
    pub = paillier.PaillierPublicKey(H(g["n"]))
    priv = paillier.PaillierPrivateKey(pub, H(g["p"]), H(g["q"]))
    rs_np = np.random.Generator(np.random.PCG64(11))
    vals = rs_np.standard_normal(37) * 10.0 ** rs_np.integers(-3, 6, 37)
    vals[5] = 0.0
    rs = [int(x) + 2 for x in rs_np.integers(0, 2 ** 62, 37)]
    host = pub.encrypt_batch(vals, r_values=rs)
    dev = pub.encrypt_batch(vals, r_values=rs, device=True)
    assert dev.on_device and not host.on_device
    assert dev.ciphertexts(False) == host.ciphertexts(False)
    w = rs_np.standard_normal(37)                        # ~half negative: exercises invert + select on the device
    w[3], w[4] = 0.0, 1.0
    assert (dev * w).ciphertexts(False) == (host * w).ciphertexts(False)
    ints = [int(x) for x in rs_np.integers(-1000, 1000, 37)]
    assert (dev * ints).ciphertexts(False) == (host * ints).ciphertexts(False)
    other_h = pub.encrypt_batch(np.arange(37, dtype=np.float64) * 0.5, r_values=rs)
    other_d = other_h.to_device()
    assert (dev + other_d).ciphertexts(False) == (host + other_h).ciphertexts(False)     # mixed exponents: alignment on device
    assert (dev + other_h).ciphertexts(False) == (host + other_h).ciphertexts(False)     # host operand is uploaded
    assert (dev + 2.5).ciphertexts(False) == (host + 2.5).ciphertexts(False)
    assert (dev - other_d).ciphertexts(False) == (host - other_h).ciphertexts(False)
    assert priv.decrypt_batch(dev) == priv.decrypt_batch(host) == vals.tolist()
    assert dev.sum().ciphertext(False) == host.sum().ciphertext(False)
    assert dev.dot(w).ciphertext(False) == host.dot(w).ciphertext(False)
    assert math.isclose(priv.decrypt(dev.dot(w)), float(np.dot(vals, w)), rel_tol=1e-9)
    assert dev[3:9].ciphertexts(False) == host[3:9].ciphertexts(False)
    assert dev[7].ciphertext(False) == host[7].ciphertext(False)
    fresh = pub.encrypt_batch(vals[:5], device=True)
    assert all(fresh._obfuscated) and priv.decrypt_batch(fresh) == vals[:5].tolist()
    un = pub.encrypt_batch([1, 2, 3], r_values=[1, 1, 1], device=True)
    before = un.ciphertexts(False)
    after = un.ciphertexts()                              # be_secure -> obfuscate_dev over the vector
    assert before != after and all(un._obfuscated) and priv.decrypt_batch(un) == [1, 2, 3]
    assert dev.to_host().ciphertexts(False) == host.ciphertexts(False)
    joined = paillier.EncryptedVector.concatenate([dev[:10], dev[10:], dev[:3]])
    assert joined.on_device and joined.ciphertexts(False) == host.ciphertexts(False) + host.ciphertexts(False)[:3]
    mixed_join = paillier.EncryptedVector.concatenate([dev[:5], host[5:]])
    assert not mixed_join.on_device and mixed_join.ciphertexts(False) == host.ciphertexts(False)


@pytest.mark.gpu
def test_decrypt_batch_pipelined_chunks():
    """decrypt_batch of a resident vector runs in chunks (download + decoding of one chunk under the kernels of the
    next): ragged chunking gives the same plaintexts as one launch, and a vector longer than the default chunk
    round-trips exactly"""
    g = load_golden(1024)
    pub = paillier.PaillierPublicKey(H(g["n"]))
    priv = paillier.PaillierPrivateKey(pub, H(g["p"]), H(g["q"]))
    rs = np.random.Generator(np.random.PCG64(8))
    vals = rs.standard_normal((1 << 16) + 4321)
    vec = pub.encrypt_batch(vals, device=True)
    assert priv.decrypt_batch(vec) == vals.tolist()
    eng = priv._get_engine()
    whole = eng.raw_decrypt_dev(vec._limbs)
    parts = list(eng.raw_decrypt_dev_chunks(vec._limbs, chunk=20000))
    assert [(lo, hi) for lo, hi, _ in parts] == [(0, 20000), (20000, 40000), (40000, 60000), (60000, len(vals))]
    assert np.array_equal(np.concatenate([p for _, _, p in parts]), whole)
    ints = rs.integers(-10 ** 9, 10 ** 9, 70000)
    assert priv.decrypt_batch(pub.encrypt_batch(ints, device=True)) == ints.tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("device", [True, False])
def test_vector_obfuscation_draws_only_for_rows_that_need_it(device):
    """EncryptedVector.obfuscate() without r_values: array-form obfuscators; rows already obfuscated keep their bits
    (r = 1 on the device path), the others change, every row still decrypts to the same value — also across the chunk
    boundaries of the pipelined device path"""
    g = load_golden(1024)
    pub = paillier.PaillierPublicKey(H(g["n"]))
    priv = paillier.PaillierPrivateKey(pub, H(g["p"]), H(g["q"]))
    count = 40000 if device else 300
    vals = np.arange(count, dtype=np.int64) - 7
    vec = pub.encrypt_batch(vals, r_values=[1] * count, device=device)       # nude ciphertexts, not obfuscated
    before = np.array(vec.to_host()._limbs)
    vec._obfuscated[::3] = True                                              # pretend every third row is done
    vec.obfuscate()
    after = np.array(vec.to_host()._limbs)
    same = (before == after).all(axis=1)
    assert same[::3].all() and not same[1::3].any() and not same[2::3].any()
    assert all(vec._obfuscated) and priv.decrypt_batch(vec) == vals.tolist()
    again = np.array(vec.obfuscate().to_host()._limbs)                        # nothing left to do
    assert np.array_equal(again, after)
    k = min(count, 5000)
    full = pub.encrypt_batch(vals[:k], r_values=[1] * k, device=device)
    nude = np.array(full.to_host()._limbs)
    assert not (np.array(full.obfuscate().to_host()._limbs) == nude).all(axis=1).any()
    assert priv.decrypt_batch(full) == vals[:k].tolist()


@pytest.mark.gpu
def test_obfuscator_pool_offline_online_split():
    """precompute_obfuscators: r^n made ahead of time; encrypt_batch / obfuscate then cost one product per element and
    give exactly raw_encrypt(m, r) for the pooled r^n; every obfuscator is consumed once; a short pool falls back"""
    from phe import _native
    g = load_golden(1024)
    pub = paillier.PaillierPublicKey(H(g["n"]))
    priv = paillier.PaillierPrivateKey(pub, H(g["p"]), H(g["q"]))
    nsq = pub.nsquare
    vals = np.arange(3000, dtype=np.int64) - 1500
    assert pub.obfuscators_available() == 0
    assert pub.precompute_obfuscators(3000) == 3000
    eng = pub._get_engine()
    first = eng.peek_obfuscators(4)
    assert all(1 < f < nsq for f in first) and len(set(first)) == 4
    vec = pub.encrypt_batch(vals[:1000], device=True)
    assert pub.obfuscators_available() == 2000 and all(vec._obfuscated) and vec.on_device
    for c, m, f in zip(vec.ciphertexts(False)[:4], vals[:4].tolist(), first):
        assert c == (1 + pub.n * (m % pub.n)) % nsq * f % nsq            # = raw_encrypt(m, r) for the pooled r^n
    assert priv.decrypt_batch(vec) == vals[:1000].tolist()
    host = pub.encrypt_batch(vals[:500].astype(np.float64))
    assert pub.obfuscators_available() == 1500 and not host.on_device
    assert priv.decrypt_batch(host) == vals[:500].astype(np.float64).tolist()
    big = pub.encrypt_batch(vals[:2000], device=True)                     # more than the pool holds: drawn on the spot
    assert pub.obfuscators_available() == 1500 and priv.decrypt_batch(big) == vals[:2000].tolist()
    un = pub.encrypt_batch(vals[:1500], r_values=[1] * 1500, device=True)
    nude = un.ciphertexts(False)
    un.obfuscate()
    assert pub.obfuscators_available() == 0 and all(un._obfuscated)
    assert all(a != b for a, b in zip(nude, un.ciphertexts(False))) and priv.decrypt_batch(un) == vals[:1500].tolist()
    pub.precompute_obfuscators(3)
    eng = pub._get_engine()                # the key pair's engine by now (the private key's; the pool moved over with it)
    assert eng is priv._get_engine()
    peek = eng.peek_obfuscators(1)[0]
    one = pub.encrypt(2.5)                                                 # scalar API: obfuscate() takes from the pool too
    nude = pub.encrypt(2.5, r_value=1)
    assert pub.obfuscators_available() == 2 and one.ciphertext(False) == nude.ciphertext(False) * peek % nsq
    assert priv.decrypt(one) == 2.5 and one._EncryptedNumber__is_obfuscated
    (one + one).ciphertext()                                               # be_secure: another pooled obfuscator
    assert pub.obfuscators_available() == 1
    pub._get_engine().take_obfuscators(1)
    pub.precompute_obfuscators(100)
    pub.precompute_obfuscators(100)
    mixed = pub.encrypt_batch(vals[:150], device=True)                     # spans two pool blocks
    assert pub.obfuscators_available() == 50 and priv.decrypt_batch(mixed) == vals[:150].tolist()
    assert len(set(mixed.ciphertexts(False))) == 150


@pytest.mark.gpu
def test_scalar_encrypt_refills_the_pool_by_the_launch(monkeypatch):
    """pub.encrypt() with an empty pool exponentiates a launch-full of obfuscators instead of one (same latency on an idle
    GPU) and serves the next calls from it; SCALAR_POOL_REFILL = 0 restores one exponentiation per call"""
    from phe import keys
    g = load_golden(1024)
    monkeypatch.setattr(keys, "SCALAR_POOL_REFILL", 64)
    pub = paillier.PaillierPublicKey(H(g["n"]))
    priv = paillier.PaillierPrivateKey(pub, H(g["p"]), H(g["q"]))
    a = pub.encrypt(1.25)
    assert pub.obfuscators_available() == 63 and a._EncryptedNumber__is_obfuscated and priv.decrypt(a) == 1.25
    cts = {pub.encrypt(7).ciphertext(False) for _ in range(63)}
    assert len(cts) == 63 and pub.obfuscators_available() == 0
    b = pub.encrypt(7, r_value=1)
    b.obfuscate()                                                          # refills as well
    assert pub.obfuscators_available() == 63 and priv.decrypt(b) == 7
    monkeypatch.setattr(keys, "SCALAR_POOL_REFILL", 0)
    pub2 = paillier.PaillierPublicKey(H(g["n"]))
    c = pub2.encrypt(-3)
    assert pub2.obfuscators_available() == 0 and priv.decrypt(c) == -3


# ---- round-2 regressions ------------------------------------------------------------------------
def test_zero_weight_against_mixed_exponents(backend):
    """A zero weight / plaintext placed against ciphertexts whose exponents are far apart used to size the aligned
    exponent rows from the non-zero rows only (Engine.shifted_limbs) and crash; the reference's chain of * and + gives
    the non-zero term.  dot, dense matvec and vec + ndarray all go through that helper."""
    g = load_golden(1024)
    pub = paillier.PaillierPublicKey(H(g["n"]))
    priv = paillier.PaillierPrivateKey(pub, H(g["p"]), H(g["q"]))
    xs = [1e10, 1e-10]
    vec = pub.encrypt_batch(np.array(xs), r_values=[5, 7])
    singles = [pub.encrypt(x, r_value=r) for x, r in zip(xs, (5, 7))]
    for w in ([0, 1], [0.0, 1.0], np.array([0.0, 1.0]), np.array([0, 3])):
        d = vec.dot(w)
        ws = w.tolist() if isinstance(w, np.ndarray) else w
        chain = singles[0] * ws[0] + singles[1] * ws[1]
        assert (d.ciphertext(False), d.exponent) == (chain.ciphertext(False), chain.exponent)
        assert priv.decrypt(d) == pytest.approx(xs[1] * ws[1], rel=1e-12)
    W = np.array([[0.0, 1.0], [2.0, 0.0], [0.0, 0.0]])
    got = vec.matvec(W)
    want = [vec.dot(row) for row in W]
    assert got.ciphertexts(False) == [x.ciphertext(False) for x in want] and got.exponents == [x.exponent for x in want]
    far = pub.encrypt_batch(np.array([1e-60, 2.0]), r_values=[5, 7])
    s = far + np.array([0.0, 1.0])
    ref = [pub.encrypt(1e-60, r_value=5) + 0.0, pub.encrypt(2.0, r_value=7) + 1.0]
    assert s.ciphertexts(False) == [x.ciphertext(False) for x in ref] and s.exponents == [x.exponent for x in ref]


def test_alignment_factor_is_bounded_by_max_int_like_the_scalar_path(backend):
    """decrease_exponent_to multiplies by BASE**delta THROUGH EncodedNumber.encode (phe/paillier.py:599,
    phe/encoding.py:194-196): a factor in (max_int, n) is a ValueError in the reference and in the scalar drop-in; the
    vector forms must raise it too, not return a ciphertext.  Such a factor exists when bit_length(n) = 1 (mod 4):
    a 253-bit modulus here (the checks run on the host, before any kernel)."""
    n = (1 << 252) + 0x1234567
    pub = paillier.PaillierPublicKey(n)
    base = phe.EncodedNumber.BASE
    d = 63
    assert pub.max_int < base ** d < pub.n and base ** (d - 1) <= pub.max_int
    one = np.zeros((2, 16), dtype=np.uint32)
    one[:, 0] = 1
    x = paillier.EncryptedNumber(pub, 1, 0)
    vec = phe.EncryptedVector(pub, one, [0, 0])
    with pytest.raises(ValueError, match="Integer needs to be within"):
        x.decrease_exponent_to(-d)
    with pytest.raises(ValueError, match="Integer needs to be within"):
        vec.decrease_exponent_to(-d)
    with pytest.raises(ValueError, match="Integer needs to be within"):
        vec.decrease_exponent_to([0, -d])
    lifted = phe.EncryptedVector(pub, one, [0, -d])
    with pytest.raises(ValueError, match="Integer needs to be within"):
        lifted.dot([1, 1])
    with pytest.raises(ValueError, match="Integer needs to be within"):
        lifted.matvec(np.array([[1, 1], [2, 3]]))
    with pytest.raises(ValueError, match="Integer needs to be within"):
        lifted + lifted[::-1]
    with pytest.raises(ValueError, match="Integer needs to be within"):
        x + paillier.EncryptedNumber(pub, 1, -d)


def test_obfuscator_pool_is_handed_out_once_across_threads():
    """Engine.take_obfuscators under concurrent callers: no pool row is ever returned twice (two ciphertexts sharing
    r^n reveal m1 - m2).  Host logic only: the pool block is a stand-in with the DeviceArray view interface."""
    import threading
    from phe import _engine

    class Block:
        cols = 8

        def __init__(self, lo, hi):
            self.lo, self.hi, self.rows = lo, hi, hi - lo

        def rows_view(self, lo, hi):
            return Block(self.lo + lo, self.lo + hi)

    eng = _engine.Engine.__new__(_engine.Engine)
    eng._lock = threading.RLock()
    eng.ct_limbs = Block.cols
    eng._pair_words = 0                                   # (no pair form on this stand-in: the pool holds ciphertext rows)
    eng._obf = _engine.ObfuscatorPool()
    eng._obf.add(Block(0, 20000))
    taken, barrier = [[] for _ in range(8)], threading.Barrier(8)

    def worker(k):
        barrier.wait()
        while True:
            part = eng.take_obfuscators(1)
            if part is None:
                return
            taken[k].append(part.lo)
    threads = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    flat = sorted(x for part in taken for x in part)
    assert flat == list(range(20000))


def test_pool_survives_the_private_engine_taking_over_under_concurrent_takers(monkeypatch):
    """pub.encrypt() in some threads while another thread creates the key pair's private engine
    for the first time.  The private engine adopts the public engine's pool OBJECT (one lock), and the creation of an
    engine is guarded: every pooled row is handed out exactly once, each key builds exactly one engine.  Host logic only:
    the engine's native context is a stand-in."""
    import threading
    import time
    from phe import _engine, keys

    class Block:
        cols = 8

        def __init__(self, lo, hi):
            self.lo, self.hi, self.rows = lo, hi, hi - lo

        def rows_view(self, lo, hi):
            return Block(self.lo + lo, self.lo + hi)

    built = []

    class StubEngine(_engine.Engine):
        def __init__(self, n, p=None, *rest, **kw):          # no native context: only what the pool logic touches
            time.sleep(0.02)                                  # a slow context creation widens the window
            self._lock = threading.RLock()
            self.n, self.ct_limbs = n, Block.cols
            self._pair_words = 0                               # (no pair form on this stand-in)
            self._obf = _engine.ObfuscatorPool()
            built.append("priv" if p is not None else "pub")

    monkeypatch.setattr(keys, "Engine", StubEngine)
    g = load_golden(256)
    pub = paillier.PaillierPublicKey(H(g["n"]))
    priv = paillier.PaillierPrivateKey(pub, H(g["p"]), H(g["q"]))
    total = 40000
    first = [None] * 4
    gate = threading.Barrier(4)

    def make(k):
        gate.wait()
        first[k] = pub._get_engine()
    ts = [threading.Thread(target=make, args=(k,)) for k in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert built == ["pub"] and all(e is first[0] for e in first)          # one engine although four threads asked at once
    first[0]._obf.add(Block(0, total))
    taken, start = [[] for _ in range(6)], threading.Barrier(7)

    def taker(k):
        start.wait()
        while True:
            part = pub._get_engine().take_obfuscators(1)          # whichever engine the key points at right now
            if part is None:
                return
            taken[k].append(part.lo)

    def owner():
        start.wait()
        time.sleep(0.005)
        priv._get_engine()
    ts = [threading.Thread(target=taker, args=(k,)) for k in range(6)] + [threading.Thread(target=owner)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert built == ["pub", "priv"]
    assert pub._get_engine() is priv._get_engine() and pub._get_engine()._obf is first[0]._obf
    flat = sorted(x for part in taken for x in part)
    assert flat == list(range(total))                                       # nothing lost, nothing handed out twice


def test_public_key_of_a_key_pair_encrypts_through_the_owner_path(backend):
    """once the private key's engine exists, the public key of the pair works through it and its encryptions take the
    key owner's CRT form (Engine.owner_encrypt) — the ciphertext bits are those of the golden fixture either way"""
    g = load_golden(1024)
    pub = paillier.PaillierPublicKey(H(g["n"]))
    before = [pub.raw_encrypt(H(e["m"]), H(e["r"])) for e in g["raw_encrypt"][:4]]          # public path
    assert not pub._get_engine().owner_encrypt()
    priv = paillier.PaillierPrivateKey(pub, H(g["p"]), H(g["q"]))
    assert priv.raw_decrypt(before[0]) == H(g["raw_encrypt"][0]["m"]) % pub.n
    eng = pub._get_engine()
    assert eng is priv._get_engine() and eng.owner_encrypt()
    enc = g["raw_encrypt"]
    assert pub.raw_encrypt_batch([H(e["m"]) for e in enc], [H(e["r"]) for e in enc]) == [H(e["c"]) for e in enc]
    assert before == [H(e["c"]) for e in enc[:4]]
    for e in g["encrypt_api"]:
        value = eval(e["value"])
        en = pub.encrypt(value, r_value=H(e["r"]))
        assert en.ciphertext(False) == H(e["c"]) and en.exponent == e["exponent"]
    xs = np.array([1.5, -2.25, 1e-3, 7.0])
    assert priv.decrypt_batch(pub.encrypt_batch(xs)) == xs.tolist()                         # fresh obfuscators


def test_obfuscation_under_a_key_pair_takes_the_owner_path_and_keeps_the_reference_bits(backend):
    """EncryptedNumber.obfuscate (phe/paillier.py:603-624: c * r^n mod n^2) through an engine that holds the private key:
    r^n comes from the CRT halves; the golden `obfuscate` vectors of the real reference must still come out"""
    g = load_golden(1024)
    pub = paillier.PaillierPublicKey(H(g["n"]))
    priv = paillier.PaillierPrivateKey(pub, H(g["p"]), H(g["q"]))
    eng = priv._get_engine()
    assert pub._get_engine() is eng and eng.owner_encrypt()
    obf = g["obfuscate"]
    got = eng.to_ints(eng.obfuscate([H(e["c_in"]) for e in obf], [H(e["r"]) for e in obf]))
    assert got == [H(e["c_out"]) for e in obf]
    vec = pub.encrypt_batch([1.5, -2.0, 3.25], r_values=[1, 1, 1])
    before = vec.ciphertexts(False)
    vec.obfuscate(r_values=[5, 7, 11])
    n2 = pub.nsquare
    assert vec.ciphertexts(False) == [c * pow(r, pub.n, n2) % n2 for c, r in zip(before, (5, 7, 11))]
    assert priv.decrypt_batch(vec) == [1.5, -2.0, 3.25]


@pytest.mark.gpu
def test_scalar_encryption_from_several_threads_never_shares_an_obfuscator():
    """pub.encrypt() from several threads on one key.  ctypes drops the GIL inside the native
    calls and the pool refills by the launch; the per-key lock must keep every r^n single-use: all obfuscators distinct,
    every value decrypts."""
    import threading
    g = load_golden(1024)
    pub = paillier.PaillierPublicKey(H(g["n"]))
    priv = paillier.PaillierPrivateKey(pub, H(g["p"]), H(g["q"]))
    per, threads = 300, 6
    out = [[None] * per for _ in range(threads)]
    errors = []

    def worker(t):
        try:
            for i in range(per):
                out[t][i] = pub.encrypt(1000 * t + i)
        except Exception as exc:                                   # surfaced below
            errors.append(exc)
    ts = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors
    n, n2 = pub.n, pub.nsquare
    seen = set()
    for t in range(threads):
        vals = priv.decrypt_batch(out[t])
        assert vals == [1000 * t + i for i in range(per)]
        for i, e in enumerate(out[t]):
            m = 1000 * t + i
            obf = e.ciphertext(False) * pow(1 + n * m, -1, n2) % n2      # = r^n of this ciphertext
            assert obf not in seen and obf != 1
            seen.add(obf)
    assert len(seen) == per * threads
