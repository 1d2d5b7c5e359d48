"""The reference's own test suites, replayed on the hot path.

tests/golden/reference_suite_trace.json.gz (made by tests/golden/gen_reference_trace.py in the build container) holds
every hot-path call — raw_encrypt, obfuscate, raw_decrypt, _raw_add, _raw_mul with operands and the value the REAL
reference returned — that /root/reference/phe/tests/paillier_test.py and math_test.py make when they run on the reference
itself, attributed to the reference test that made it (the 17-bit known-answer key and 2048-bit keys, both _raw_mul branches, plaintexts at
and above n, obfuscators 1 and random).  The GPU box has no /root/reference; here the trace is replayed

  * `-m gpu`: through libphe_hip.so (the host mirror Engine -> ctypes -> C-ABI -> HIP kernels), one pytest case per
    reference test, every result compared bit for bit;
  * CPU: through both restatements in oracle/ (pins the oracle to the suite) and, for a bounded subset, through the device
    headers on the wave emulator.
"""
import gzip
import json
import os
import sys

import pytest

from conftest import GOLDEN, PKG

if PKG not in sys.path:
    sys.path.insert(0, PKG)

with gzip.open(os.path.join(GOLDEN, "reference_suite_trace.json.gz"), "rb") as _f:
    TRACE = json.loads(_f.read().decode())

KEYS = [{k: (int(v, 16) if v is not None else None) for k, v in key.items()} for key in TRACE["keys"]]
OPS_BY_TEST = {}
for _row in TRACE["ops"]:
    OPS_BY_TEST.setdefault(_row[0], []).append((_row[1], _row[2], [int(v, 16) for v in _row[3:]]))
TEST_IDS = sorted(OPS_BY_TEST)


def test_trace_covers_the_suite():
    assert TRACE["tests_run"] >= 188 and len(TEST_IDS) == TRACE["tests_with_hot_path_calls"] >= 85
    kinds = {}
    for ops in OPS_BY_TEST.values():
        for op, k, vals in ops:
            kinds[op] = kinds.get(op, 0) + 1
    assert set(kinds) == {"enc", "obf", "dec", "add", "mul"} and min(kinds.values()) >= 50
    bits = sorted({key["n"].bit_length() for key in KEYS})
    assert bits[0] <= 32 and 2048 in bits                                          # the 17-bit KAT key and the default size
    # both _raw_mul branches occur (phe/paillier.py:745-751)
    neg = [vals[1] >= KEYS[k]["n"] - (KEYS[k]["n"] // 3 - 1) for ops in OPS_BY_TEST.values() for op, k, vals in ops if op == "mul"]
    assert any(neg) and not all(neg)
    assert all(key["p"] for i, key in enumerate(KEYS) if any(op == "dec" and k == i for ops in OPS_BY_TEST.values() for op, k, _ in ops))


def test_oracles_reproduce_the_trace(c_oracle):
    """both CPU restatements (oracle/paillier_oracle.py on CPython ints, oracle/paillier_oracle.c on libgmp) return what
    the reference returned for every recorded call"""
    from oracle.paillier_oracle import PyPrivate, PyPublic, int_to_limbs, ints_to_limbs, limbs_to_ints
    import numpy as np
    pubs = [PyPublic(key["n"]) for key in KEYS]
    privs = [PyPrivate(pub, key["p"], key["q"]) if key["p"] else None for pub, key in zip(pubs, KEYS)]
    count = 0
    for t in TEST_IDS:
        for op, k, v in OPS_BY_TEST[t]:
            pub, n = pubs[k], KEYS[k]["n"]
            s1 = max(1, (n.bit_length() + 31) // 32)
            n_arr = int_to_limbs(n, s1)
            row = lambda x, limbs: ints_to_limbs([x], limbs)
            if op == "enc":
                assert pub.raw_encrypt(v[0], v[1]) == v[2]
                got = c_oracle.encrypt(n_arr, row(v[0] % n, s1), row(v[1], s1))
            elif op == "obf":
                assert pub.obfuscate(v[0], v[1]) == v[2]
                got = c_oracle.obfuscate(n_arr, row(v[0] % (n * n), 2 * s1), row(v[1], s1))
            elif op == "dec":
                assert privs[k].raw_decrypt(v[0]) == v[1]
                pq = max(1, (max(KEYS[k]["p"], KEYS[k]["q"]).bit_length() + 31) // 32)
                got = c_oracle.decrypt(n_arr, int_to_limbs(privs[k].p, pq), int_to_limbs(privs[k].q, pq), row(v[0] % (n * n), 2 * s1))
            elif op == "add":
                assert pub.raw_add(v[0], v[1]) == v[2]
                got = c_oracle.add(n_arr, row(v[0] % (n * n), 2 * s1), row(v[1] % (n * n), 2 * s1))
            else:
                assert pub.raw_mul(v[0], v[1]) == v[2]
                got = c_oracle.mul(n_arr, row(v[0] % (n * n), 2 * s1), row(v[1], s1))
            assert limbs_to_ints(np.asarray(got))[0] == v[-1], (TRACE["tests"][t], op)
            count += 1
    assert count == len(TRACE["ops"])


_ENGINES = {}


def _engine(k):
    from phe._engine import Engine
    if k not in _ENGINES:
        key = KEYS[k]
        if key["p"]:
            from phe import paillier
            priv = paillier.PaillierPrivateKey(paillier.PaillierPublicKey(key["n"]), key["p"], key["q"])
            _ENGINES[k] = Engine(key["n"], priv.p, priv.q, priv.hp, priv.hq, priv.p_inverse)
        else:
            _ENGINES[k] = Engine(key["n"])
    return _ENGINES[k]


def _replay(ops):
    """the recorded calls of one reference test through Engine (the host mirror of the five hot functions)"""
    for op, k, v in ops:
        eng = _engine(k)
        if op == "enc":
            got = eng.to_ints(eng.raw_encrypt([v[0]], [v[1]]))[0]
        elif op == "obf":
            got = eng.to_ints(eng.obfuscate([v[0]], [v[1]]))[0]
        elif op == "dec":
            got = eng.to_ints(eng.raw_decrypt([v[0]]))[0]
        elif op == "add":
            got = eng.to_ints(eng.raw_add([v[0]], [v[1]]))[0]
        else:
            got = eng.to_ints(eng.raw_mul([v[0]], [v[1]]))[0]
        assert got == v[-1], (op, KEYS[k]["n"].bit_length())


@pytest.mark.gpu
@pytest.mark.parametrize("t", TEST_IDS, ids=[TRACE["tests"][t].replace("phe.tests.", "") for t in TEST_IDS])
def test_reference_test_replayed_on_the_hip_path(t):
    """one case per reference test that touches the hot path: same operands -> same bits from libphe_hip.so"""
    from phe import _native
    assert _native.Context.__module__ == "phe._native", "the real C-ABI context, not the emulator stand-in"
    _replay(OPS_BY_TEST[t])


@pytest.mark.gpu
def test_whole_trace_batched_per_key_on_the_hip_path():
    """the same calls again, this time as ONE batch per (key, operation): the shape the batched API is made for"""
    per = {}
    for ops in OPS_BY_TEST.values():
        for op, k, v in ops:
            per.setdefault((k, op), []).append(v)
    total = 0
    for (k, op), rows in sorted(per.items()):
        eng = _engine(k)
        cols = list(zip(*rows))
        if op == "enc":
            got = eng.raw_encrypt(list(cols[0]), list(cols[1]))
        elif op == "obf":
            got = eng.obfuscate(list(cols[0]), list(cols[1]))
        elif op == "dec":
            got = eng.raw_decrypt(list(cols[0]))
        elif op == "add":
            got = eng.raw_add(list(cols[0]), list(cols[1]))
        else:
            got = eng.raw_mul(list(cols[0]), list(cols[1]))
        assert eng.to_ints(got) == list(cols[-1]), (op, KEYS[k]["n"].bit_length())
        total += len(rows)
    assert total == len(TRACE["ops"])


def test_bounded_subset_on_the_wave_emulator(monkeypatch):
    """CPU stand-in for the GPU replay: the device headers on the wave emulator, for every key below 1100 bits and the
    first few calls of each kind under the larger keys (the emulator is ~10^4 times slower than the GPU)"""
    import emu_backend
    emu_backend.install(monkeypatch)
    _ENGINES.clear()
    budget = {}
    try:
        for t in TEST_IDS:
            keep = []
            for op, k, v in OPS_BY_TEST[t]:
                bits = KEYS[k]["n"].bit_length()
                if bits > 1100:
                    used = budget.get((op, bits > 2100), 0)
                    if used >= (2 if bits > 2100 else 4):
                        continue
                    budget[(op, bits > 2100)] = used + 1
                keep.append((op, k, v))
            _replay(keep)
    finally:
        _ENGINES.clear()
