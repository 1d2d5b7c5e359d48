"""Single-process fan-out over several devices behind the drop-in API (phe/fleet.py).

  * CPU (default run): two EMULATOR contexts stand in for two devices (PHE_HIP_DEVICES=0,1 with tests/emu_backend.EmuContext):
    the shards, the worker threads, the per-device engines / pools and the order of the results are the product's own code;
  * GPU (-m gpu): two contexts on device 0 (PHE_HIP_DEVICES=0,0 — each context has its own streams, tables and pool), results
    bit-identical to the one-context result, the obfuscator pool never shared between the two.
Nothing here reads /root/reference."""
import os
import sys
import threading

import numpy as np
import pytest

from conftest import PKG, load_golden

if PKG not in sys.path:
    sys.path.insert(0, PKG)

import phe as paillier  # noqa: E402
from phe import fleet  # noqa: E402


def H(x):
    return int(x, 16)


def test_device_list_parsing(monkeypatch):
    monkeypatch.delenv("PHE_HIP_DEVICES", raising=False)
    assert fleet.configured_devices() is None
    monkeypatch.setenv("PHE_HIP_DEVICES", "3")
    assert fleet.configured_devices() is None                       # one device: no fan-out
    monkeypatch.setenv("PHE_HIP_DEVICES", "2, 3,5")
    assert fleet.configured_devices() == [2, 3, 5]
    monkeypatch.setenv("PHE_HIP_DEVICES", "0,0")
    assert fleet.configured_devices() == [0, 0]                     # two contexts on one GPU
    monkeypatch.setenv("PHE_HIP_DEVICES", "zero")
    with pytest.raises(ValueError):
        fleet.configured_devices()
    from phe import _engine
    monkeypatch.setenv("PHE_HIP_DEVICES", "2,3")
    monkeypatch.delenv("PHE_HIP_DEVICE", raising=False)
    assert _engine.default_device() == 2                            # the ordinary engine sits on the fleet's first device


def test_shards_are_the_multi_process_cut():
    fl = fleet.Fleet(primary=object(), make_engine=lambda d: object(), devices=[0, 1, 2])
    assert fl.shards(10, min_rows=1) == [(0, 4), (4, 7), (7, 10)]     # sharding.shard_bounds: contiguous, balanced
    assert fl.shards(10, min_rows=4) == [(0, 5), (5, 10)]             # not worth the third device
    assert fl.shards(3, min_rows=4) == [(0, 3)]                       # no fan-out
    seen = []
    out = fl.run(10, lambda eng, lo, hi: (seen.append(threading.current_thread().name), (lo, hi))[1], min_rows=1)
    assert out == [(0, 4), (4, 7), (7, 10)] and all(name.startswith("phe-fleet") for name in seen)
    with pytest.raises(ZeroDivisionError):                            # a shard's error reaches the caller once all are done
        fl.run(10, lambda eng, lo, hi: 1 // (lo - 4), min_rows=1)


@pytest.fixture(params=["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def two_contexts(request, monkeypatch):
    if request.param == "emu":
        import emu_backend
        emu_backend.install(monkeypatch)
        monkeypatch.setenv("PHE_HIP_DEVICES", "0,1")
        key_bits, rows = 256, 14
    else:
        monkeypatch.setenv("PHE_HIP_DEVICES", "0,0")
        key_bits, rows = 2048, 6000
    monkeypatch.setattr(fleet, "MIN_ROWS_PER_DEVICE", 4 if request.param == "emu" else 2048)
    return request.param, key_bits, rows


def test_encrypt_decrypt_add_mul_fan_out_over_two_contexts(two_contexts, monkeypatch):
    kind, key_bits, rows = two_contexts
    g = load_golden(key_bits)
    pub = paillier.PaillierPublicKey(H(g["n"]))
    priv = paillier.PaillierPrivateKey(pub, H(g["p"]), H(g["q"]))
    rng = np.random.RandomState(5)
    x = rng.uniform(-1e3, 1e3, size=rows)
    k = rng.randint(-50, 50, size=rows).astype(np.int64)
    fl = priv._get_fleet()
    assert fl is not None and len(fl) == 2 and pub._get_fleet() is fl            # a key pair shares one fleet
    calls = {}
    for i, eng in enumerate(fl.engines()):                                       # count the rows every engine really worked on
        for name in ("raw_encrypt_fresh", "raw_encrypt", "raw_decrypt", "raw_add", "raw_mul_signed"):
            if hasattr(eng, name):
                def counted(*a, _f=getattr(eng, name), _key=(i, name), **kw):
                    first = a[0]
                    calls[_key] = calls.get(_key, 0) + (first.shape[0] if hasattr(first, "shape") else len(first))
                    return _f(*a, **kw)
                monkeypatch.setattr(eng, name, counted)
    assert fl.engine(0) is priv._get_engine() and fl.engine(1) is not fl.engine(0)
    assert fl.engine(0)._obf is not fl.engine(1)._obf                            # a pool per device, never shared
    vec = pub.encrypt_batch(x)
    assert not vec.on_device and len(vec) == rows
    enc_rows = [sum(v for (i, name), v in calls.items() if i == dev and name.startswith("raw_encrypt")) for dev in (0, 1)]
    assert enc_rows == [rows - rows // 2, rows // 2], (enc_rows, calls)          # contiguous halves, one per context
    got = priv.decrypt_batch(vec)
    assert np.allclose(got, x, rtol=0, atol=1e-9)
    assert [calls.get((dev, "raw_decrypt"), 0) for dev in (0, 1)] == [rows - rows // 2, rows // 2]
    # the same ciphertexts through ONE context decrypt to the same numbers, and a one-context encryption decrypts on the fleet
    monkeypatch.delenv("PHE_HIP_DEVICES")
    pub1 = paillier.PaillierPublicKey(H(g["n"]))
    priv1 = paillier.PaillierPrivateKey(pub1, H(g["p"]), H(g["q"]))
    assert priv1._get_fleet() is None
    same = paillier.EncryptedVector(pub1, vec._limbs.copy(), vec.exponent_array.copy())
    assert priv1.decrypt_batch(same) == got
    # homomorphic operators on host vectors: bit-identical to the one-context result (no randomness in them)
    r = [int(v) for v in rng.randint(1, 1 << 30, size=rows)]
    fixed = pub1.encrypt_batch(x, r_values=r)                                    # explicit r: deterministic ciphertexts
    fixed_fleet = paillier.EncryptedVector(pub, fixed._limbs.copy(), fixed.exponent_array.copy())
    monkeypatch.setenv("PHE_HIP_DEVICES", "0,1" if kind == "emu" else "0,0")
    assert np.array_equal(pub.encrypt_batch(x, r_values=r)._limbs, fixed._limbs)          # fan-out with given r: same bits
    prod_fleet, prod_one = fixed_fleet * k, fixed * k
    assert np.array_equal(prod_fleet._limbs, prod_one._limbs)
    assert [calls.get((dev, "raw_mul_signed"), 0) for dev in (0, 1)] == [rows - rows // 2, rows // 2]
    monkeypatch.setattr(fleet, "MIN_ROWS_PER_DEVICE", 1)                          # (the sum's threshold is 8 x this)
    sum_fleet, sum_one = fixed_fleet + prod_fleet, fixed + prod_one
    assert np.array_equal(sum_fleet._limbs, sum_one._limbs)
    if rows >= 16:
        assert calls.get((1, "raw_add"), 0) == rows // 2
    want = x + x * k
    assert np.allclose(priv.decrypt_batch(sum_fleet), want, rtol=1e-12, atol=1e-6)


@pytest.mark.gpu
def test_large_host_additions_take_the_tile_kernel_on_both_contexts_at_once(monkeypatch):
    """`vec + vec` on host vectors of 40,000 rows through two contexts: each shard (20,000 rows) is one launch of k_mulmod_tile
    (csrc/mul_tile.h) on its own stream from its own worker thread — the kernels' dynamic-LDS limit is raised per launch, per
    device — and the sum is bit-identical to the one-context sum and decrypts to x + y."""
    g = load_golden(2048)
    rows = 40000
    rng = np.random.RandomState(11)
    x, y = rng.randint(-10 ** 6, 10 ** 6, size=rows), rng.randint(-10 ** 6, 10 ** 6, size=rows)
    monkeypatch.delenv("PHE_HIP_DEVICES", raising=False)
    pub1 = paillier.PaillierPublicKey(H(g["n"]))
    a1, b1 = pub1.encrypt_batch(x), pub1.encrypt_batch(y)
    one = a1 + b1
    monkeypatch.setenv("PHE_HIP_DEVICES", "0,0")
    pub = paillier.PaillierPublicKey(H(g["n"]))
    priv = paillier.PaillierPrivateKey(pub, H(g["p"]), H(g["q"]))
    fl = priv._get_fleet()
    assert fl is not None and len(fl) == 2
    a = paillier.EncryptedVector(pub, a1._limbs.copy(), a1.exponent_array.copy())
    b = paillier.EncryptedVector(pub, b1._limbs.copy(), b1.exponent_array.copy())
    both = a + b
    assert np.array_equal(both._limbs, one._limbs)
    for eng in fl.engines():
        path = eng.ctx.last_launch()["path"]
        assert path & eng.ctx.PATH_TILE_MUL, path                                 # 20,000 rows per context: by tiles
    assert priv.decrypt_batch(both) == (x + y).tolist()


@pytest.mark.gpu
def test_resident_shards_and_per_device_pools(monkeypatch):
    """encrypt_batch_sharded: one resident vector per context, each worked on by ITS engine; obfuscators made ahead of time are
    split over the contexts' pools and every one is used once"""
    monkeypatch.setenv("PHE_HIP_DEVICES", "0,0")
    g = load_golden(2048)
    pub = paillier.PaillierPublicKey(H(g["n"]))
    priv = paillier.PaillierPrivateKey(pub, H(g["p"]), H(g["q"]))
    fl = priv._get_fleet()
    rows = 3001
    assert pub.precompute_obfuscators(rows + 9, sharded=True) == rows + 9      # cut like encrypt_batch_sharded: a part per context
    per_pool = [eng.obfuscators_available() for eng in fl.engines()]
    assert per_pool == [1505, 1505] == pub.obfuscators_available(per_device=True) and fl.engine(0)._obf is not fl.engine(1)._obf
    peek = [set(eng.peek_obfuscators(50)) for eng in fl.engines()]
    assert not (peek[0] & peek[1])                                                # different r^n in the two pools
    x = np.linspace(-7.5, 9.25, rows)
    parts = pub.encrypt_batch_sharded(x)
    assert [len(p) for p in parts] == [1501, 1500] and all(p.on_device for p in parts)
    assert [p._eng() is fl.engine(i) for i, p in enumerate(parts)] == [True, True]
    assert [eng.obfuscators_available() for eng in fl.engines()] == [4, 5]        # consumed from each device's own pool
    doubled = [p + p for p in parts]                                               # resident ops stay with the owning engine
    assert np.allclose(priv.decrypt_batch(doubled), 2 * x, rtol=0, atol=1e-9)
    mixed = parts[0][:1500] + parts[1]                                             # operands of two contexts: one comes over
    assert np.allclose(priv.decrypt_batch(mixed), x[:1500] + x[1501:], rtol=0, atol=1e-9)
    pub.discard_obfuscators()
    assert pub.obfuscators_available() == 0


def test_three_contexts_uneven_shards_through_encrypt_add_decrypt(monkeypatch):
    """no multi-GPU hardware has run this path, so the plumbing must not be able to fail there — three
    emulator contexts (distinct device ids), 26 rows cut 9 + 9 + 8, encrypt_batch -> `+` -> `*` -> decrypt_batch, a pool and a
    lock per context, results in shard order"""
    import emu_backend
    emu_backend.install(monkeypatch)
    monkeypatch.setenv("PHE_HIP_DEVICES", "0,1,2")
    monkeypatch.setattr(fleet, "MIN_ROWS_PER_DEVICE", 2)
    g = load_golden(256)
    pub = paillier.PaillierPublicKey(H(g["n"]))
    priv = paillier.PaillierPrivateKey(pub, H(g["p"]), H(g["q"]))
    fl = priv._get_fleet()
    assert len(fl) == 3 and [e.device for e in fl.engines()] == [0, 1, 2]
    assert len({id(e._obf) for e in fl.engines()}) == 3 and len({id(e.ctx) for e in fl.engines()}) == 3
    assert fl.shards(26) == [(0, 9), (9, 18), (18, 26)]
    rows_seen = {}
    for e in fl.engines():
        for name in ("raw_encrypt", "raw_decrypt", "raw_add"):
            def counted(*a, _f=getattr(e, name), _key=(e.device, name), **kw):
                first = a[0]
                rows_seen[_key] = rows_seen.get(_key, 0) + (first.shape[0] if hasattr(first, "shape") else len(first))
                return _f(*a, **kw)
            monkeypatch.setattr(e, name, counted)
    rng = np.random.RandomState(3)
    x, y = rng.uniform(-50, 50, size=26), rng.uniform(-50, 50, size=26)
    a, b = pub.encrypt_batch(x), pub.encrypt_batch(y)
    assert [rows_seen[(d, "raw_encrypt")] for d in (0, 1, 2)] == [18, 18, 16]
    monkeypatch.setattr(fleet, "MIN_ROWS_PER_DEVICE", 1)                    # (the sum's threshold is 8 x this)
    total = a + b
    assert [rows_seen.get((d, "raw_add"), 0) for d in (0, 1, 2)] == [9, 9, 8]
    monkeypatch.setattr(fleet, "MIN_ROWS_PER_DEVICE", 2)
    got = priv.decrypt_batch(total)
    assert [rows_seen[(d, "raw_decrypt")] for d in (0, 1, 2)] == [9, 9, 8]
    assert np.allclose(got, x + y, rtol=0, atol=1e-9)


class _Rows:  # what _engine_for looks at of a resident array: the native context it was made in
    def __init__(self, ctx):
        self.ctx = ctx


def test_private_engine_takes_over_a_public_fleet_device_for_device(monkeypatch):
    """`pub.encrypt_batch_sharded(x)` BEFORE the first private-key call builds a public-only fleet; the
    key pair's engine then replaces the public one.  The resident parts made on devices 1..k must resolve to engines ON those
    devices that hold the private key (never to the device-0 engine: nothing enables peer access), and the per-device
    obfuscator pools must travel with their devices."""
    import emu_backend
    emu_backend.install(monkeypatch)
    monkeypatch.setenv("PHE_HIP_DEVICES", "0,1,2")
    g = load_golden(256)
    pub = paillier.PaillierPublicKey(H(g["n"]))
    priv = paillier.PaillierPrivateKey(pub, H(g["p"]), H(g["q"]))
    old = pub._get_fleet()                                               # public-only: what encrypt_batch_sharded would use
    old_engines = old.engines()
    assert not any(e.ctx.has_private for e in old_engines)
    parts = [_Rows(e.ctx) for e in old_engines]                          # "resident vectors", one per device
    assert [pub._engine_for(p) for p in parts] == old_engines
    priv._get_engine()                                                   # the key pair's engine takes over
    new = pub._get_fleet()
    assert new is not old and new is priv._get_fleet() and new.devices == [0, 1, 2]
    # the successors on devices 1.. are made on FIRST USE (the take-over runs under the key's engine lock), the
    # replaced fleet's worker threads are told to exit, and of a retired engine only its native context is kept
    assert len(new.made()) == 1 and sorted(new._inherit) == [1, 2]
    assert old._pool._shutdown and all(not hasattr(c, "ctx") for c, _ in new._retired)
    assert sorted(k for _, k in new._retired) == [0, 1, 2] and [c for c, _ in new._retired] == [e.ctx for e in old_engines]
    homes = [pub._engine_for(p) for p in parts]
    assert not new._inherit                                              # adopted when the successors were made
    assert [e.device for e in homes] == [0, 1, 2] and all(e.ctx.has_private for e in homes)
    assert homes == new.engines() and homes[0] is priv._get_engine()
    assert [e._obf for e in homes] == [e._obf for e in old_engines]      # each pool stayed with its device
    # a second take-over (another PaillierPrivateKey object for the same public key) keeps resolving the first fleet's contexts
    stranger = _Rows(type("Ctx", (), {"device": 7})())
    with pytest.raises(RuntimeError, match="device 7"):
        pub._engine_for(stranger)
    same_gpu = _Rows(type("Ctx", (), {"device": 2})())                   # an unknown context of a served GPU: pointers are valid there
    assert pub._engine_for(same_gpu) is homes[2]


@pytest.mark.gpu
def test_shards_made_before_the_private_engine_exists_decrypt_where_they_live(monkeypatch):
    """the documented flow, in the order ADVICE round 4 found broken: generate -> pub.encrypt_batch_sharded -> priv.decrypt_batch
    (two contexts on device 0 stand in for two GPUs; the second part's home must be the successor of the context that made it)"""
    monkeypatch.setenv("PHE_HIP_DEVICES", "0,0")
    g = load_golden(2048)
    pub = paillier.PaillierPublicKey(H(g["n"]))
    priv = paillier.PaillierPrivateKey(pub, H(g["p"]), H(g["q"]))
    x = np.linspace(-3.5, 8.25, 4100)
    pub.precompute_obfuscators(64, sharded=True)
    parts = pub.encrypt_batch_sharded(x)                                 # public-only fleet
    old = pub._fleet
    assert len(parts) == 2 and all(p.on_device for p in parts) and not any(e.ctx.has_private for e in old.made())
    pools = [e._obf for e in old.engines()]
    got = priv.decrypt_batch(parts)                                      # builds the key pair's engines
    assert np.allclose(got, x, rtol=0, atol=1e-9)
    new = pub._fleet
    assert new is not old and [e._obf for e in new.engines()] == pools
    assert [parts[i]._eng() is new.engine(i) for i in (0, 1)] == [True, True]
    assert np.allclose(priv.decrypt_batch([p + p for p in parts]), 2 * x, rtol=0, atol=1e-9)
