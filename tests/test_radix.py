"""The decimal wire format in batch form (csrc/radix_conv.h, kernels_radix.hip, include/phe_hip.h
phe_hip_to_decimal / phe_hip_from_decimal).

The reference serialises every ciphertext as str(int) and reads it back with int(str)
(docs/serialisation.rst:24-43; phe/command_line.py:120-131, :267-276), so the check is exactly that: the digits must
be Python's.  CPU run: the per-number routines compiled for the host (tests/emu); -m gpu: the kernels through the
C-ABI, host and device-resident inputs, ragged batches, every limb width the key sizes produce."""
import json
import random
import sys

import numpy as np
import pytest

from conftest import PKG, load_golden

if PKG not in sys.path:
    sys.path.insert(0, PKG)

from oracle.paillier_oracle import ints_to_limbs, limbs_to_ints  # noqa: E402


def H(x):
    return int(x, 16)


def _numbers(rng, words, count):
    top = 1 << (32 * words)
    xs = [0, 1, 9, 10, 999999999, 10 ** 9, 10 ** 9 + 1, 10 ** 18 - 1, 10 ** 18, top - 1, top >> 1, (top >> 1) - 1]
    xs += [rng.getrandbits(rng.randrange(1, 32 * words + 1)) for _ in range(count)]
    return [x % top for x in xs]


def _strings(digits):
    return [bytes(r).decode().lstrip("0") or "0" for r in digits]


def _check_backend(to_decimal, from_decimal, width_of, words, xs):
    limbs = ints_to_limbs(xs, words)
    digits = to_decimal(limbs)
    assert digits.shape == (len(xs), width_of(words)) and digits.dtype == np.uint8
    assert _strings(digits) == [str(x) for x in xs]                       # str(int), digit for digit
    assert limbs_to_ints(from_decimal(digits, words)) == xs               # int(str)
    padded = np.concatenate([np.full((len(xs), 7), ord("0"), np.uint8), digits], axis=1)
    assert limbs_to_ints(from_decimal(padded, words)) == xs               # any amount of '0' padding
    short = max(len(str(x)) for x in xs[:9])                              # narrow field holding small numbers only
    raw = b"".join(str(x).encode().rjust(short, b"0") for x in xs[:9])
    assert limbs_to_ints(from_decimal(np.frombuffer(raw, np.uint8).reshape(9, short), words)) == xs[:9]


@pytest.fixture(scope="module")
def emu():
    from emu_lib import Emu
    return Emu()


@pytest.mark.parametrize("words", [1, 2, 3, 16, 64, 128, 192, 256])
def test_conversion_routines_on_the_host(emu, words):
    rng = random.Random(words)
    _check_backend(emu.to_decimal, emu.from_decimal, emu.L.emu_decimal_width, words, _numbers(rng, words, 40))


def test_conversion_errors_on_the_host(emu):
    digits = emu.to_decimal(ints_to_limbs([5, 7, 11], 2))
    bad = digits.copy()
    bad[1, 3] = ord("x")
    bad[2, 0] = ord("-")
    with pytest.raises(ValueError) as ei:                                  # int("12x") raises ValueError
        emu.from_decimal(bad, 2)
    assert ei.value.bad_index == 1
    big = np.frombuffer(str(1 << 64).rjust(25, "0").encode(), np.uint8).reshape(1, -1)
    with pytest.raises(ValueError):
        emu.from_decimal(big, 2)
    assert limbs_to_ints(emu.from_decimal(big, 3)) == [1 << 64]
    with pytest.raises(ValueError):
        emu.to_decimal(ints_to_limbs([12345678901], 2), width=10)


def test_json_vector_format_is_what_json_dumps_writes(monkeypatch):
    import emu_backend
    emu_backend.install(monkeypatch)
    from phe import paillier
    g = load_golden(256)
    pub = paillier.PaillierPublicKey(H(g["n"]))
    priv = paillier.PaillierPrivateKey(pub, H(g["p"]), H(g["q"]))
    vals = [0.5, -1.25, 3.0, 1e-3, 12345]
    vec = pub.encrypt_batch(vals, r_values=[H(e["r"]) for e in g["raw_encrypt"][:5]])
    text = vec.to_json(be_secure=False)
    want = json.dumps({"public_key": {"n": pub.n},
                       "values": [[str(c), e] for c, e in zip(vec.ciphertexts(False), vec.exponents)]})
    assert text == want                                                    # docs/serialisation.rst:24-43, byte for byte
    back = paillier.EncryptedVector.from_json(text)
    assert back.ciphertexts(False) == vec.ciphertexts(False) and back.exponents == vec.exponents
    assert priv.decrypt_batch(back) == vals
    doc = json.loads(text)
    doc["values"][0][0] = int(doc["values"][0][0])                         # a JSON number instead of a string also loads
    assert paillier.EncryptedVector.from_json(json.dumps(doc)).ciphertexts(False) == vec.ciphertexts(False)
    doc["values"][1][0] = "12a4"
    with pytest.raises(ValueError):
        paillier.EncryptedVector.from_json(json.dumps(doc))
    assert paillier.EncryptedVector.from_json(vec[:0].to_json(False)).ciphertexts(False) == []


# ---- GPU -------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ctx():
    from phe import _native
    assert _native.device_count() >= 1
    g = load_golden(1024)
    return _native.Context(H(g["n"]), n_limbs=32)


@pytest.mark.gpu
@pytest.mark.parametrize("words", [1, 2, 16, 64, 128, 192, 256])
def test_kernels_against_python_str_and_int(ctx, words):
    """every limb width the key sizes produce (64 .. 256 words = 1024 .. 4096-bit keys; 256 words needs the large LDS
    tile), batches around the 64-number workgroup and large enough for several waves per CU"""
    rng = random.Random(words)
    xs = _numbers(rng, words, 53 if words > 64 else 1500)
    _check_backend(ctx.to_decimal, ctx.from_decimal, ctx.decimal_width, words, xs)
    for count in (1, 63, 64, 65):
        _check_backend(ctx.to_decimal, ctx.from_decimal, ctx.decimal_width, words, _numbers(rng, words, count)[-count:] + [0] * 9)


@pytest.mark.gpu
def test_kernel_errors(ctx):
    digits = ctx.to_decimal(ints_to_limbs([5, 7, 11, 13], 4))
    bad = digits.copy()
    bad[2, 5] = ord(" ")
    bad[3, 1] = ord("x")
    with pytest.raises(ValueError) as ei:
        ctx.from_decimal(bad, 4)
    assert ei.value.bad_index == 2
    big = np.frombuffer(str(1 << 128).rjust(50, "0").encode(), np.uint8).reshape(1, -1)
    with pytest.raises(ValueError):
        ctx.from_decimal(big, 4)
    assert limbs_to_ints(ctx.from_decimal(big, 5)) == [1 << 128]
    assert ctx.to_decimal(np.zeros((0, 4), np.uint32)).shape[0] == 0


@pytest.mark.gpu
def test_json_of_a_resident_vector():
    from phe import paillier
    g = load_golden(2048)
    pub = paillier.PaillierPublicKey(H(g["n"]))
    priv = paillier.PaillierPrivateKey(pub, H(g["p"]), H(g["q"]))
    rs = np.random.Generator(np.random.PCG64(3))
    vals = rs.standard_normal(3000)
    dev = pub.encrypt_batch(vals, device=True)
    text = dev.to_json()                                                   # be_secure: already obfuscated, digits from HBM
    want = json.dumps({"public_key": {"n": pub.n},
                       "values": [[str(c), e] for c, e in zip(dev.ciphertexts(False), dev.exponents)]})
    assert text == want
    back = paillier.EncryptedVector.from_json(text, device=True)
    assert back.on_device and priv.decrypt_batch(back) == vals.tolist()
    assert dev.to_host().to_json(False) == want
