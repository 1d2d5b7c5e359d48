#!/usr/bin/env python3
"""The call shape of BASELINE.json configs[4]'s file on the GPU, through the drop-in's SCALAR API.

/root/reference/examples/federated_learning_with_encryption.py is not on the GPU box, so the file itself cannot run there
(tests/test_reference_example_verbatim.py runs it verbatim in the build container).  What it does to the backend is, per run
(5 hospitals, 50 rounds, 10 features + intercept = 11 gradient components; :122-133, :213-225):

    2,750 x public_key.encrypt(float)          one at a time      (:123  [public_key.encrypt(i) for i in gradient])
    2,200 x EncryptedNumber + EncryptedNumber  one at a time      (:131  sum_encrypted_vectors)
      550 x private_key.decrypt(enc)           one at a time      (:156  decrypt_aggregate)

This tool makes exactly those calls, in that order and interleaving, on gradient-sized floats, with the key size the file asks
for (1024 bits, :257) and with 2048 bits, and reports seconds per phase and per call — the scalar-loop figure VERDICT round 4
(missing 5) asked for — next to the batched form of the same protocol (examples/federated_learning_batched.py).

    python tools/federated_scalar_shape.py > profiles/rNN_federated_scalar_shape.json      (GPU box)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "python-paillier_amd"), os.path.join(ROOT, "examples"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402

N_CLIENTS, N_ROUNDS, N_COMPONENTS = 5, 50, 11


def run(key_bits, phe):
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "paillier_%d.json" % key_bits)))
    H = lambda k: int(g[k], 16)
    pub = phe.PaillierPublicKey(H("n"))
    priv = phe.PaillierPrivateKey(pub, H("p"), H("q"))
    rng = np.random.RandomState(key_bits)
    t = {"encrypt": 0.0, "add": 0.0, "decrypt": 0.0}
    calls = {"encrypt": 0, "add": 0, "decrypt": 0}
    pub.encrypt(0.5), priv.decrypt(pub.encrypt(0.25))                       # contexts, pools and kernels warm (as after round 1)
    worst = 0.0
    t_all = time.perf_counter()
    for _ in range(N_ROUNDS):
        grads = rng.uniform(-30, 30, size=(N_CLIENTS, N_COMPONENTS))
        t0 = time.perf_counter()
        agg = [pub.encrypt(float(x)) for x in grads[0]]                   # :123
        t["encrypt"] += time.perf_counter() - t0
        calls["encrypt"] += N_COMPONENTS
        for c in range(1, N_CLIENTS):
            t0 = time.perf_counter()
            enc = [pub.encrypt(float(x)) for x in grads[c]]
            t1 = time.perf_counter()
            agg = [a + b for a, b in zip(agg, enc)]                          # :131
            t2 = time.perf_counter()
            t["encrypt"] += t1 - t0
            t["add"] += t2 - t1
            calls["encrypt"] += N_COMPONENTS
            calls["add"] += N_COMPONENTS
        t0 = time.perf_counter()
        got = [priv.decrypt(a) for a in agg]                                 # :156
        t["decrypt"] += time.perf_counter() - t0
        calls["decrypt"] += N_COMPONENTS
        worst = max(worst, float(np.max(np.abs(np.array(got) - grads.sum(axis=0)))))
    total = time.perf_counter() - t_all
    return {"key_bits": key_bits, "seconds_total": total, "calls": calls,
            "seconds": t, "ms_per_call": {k: 1e3 * t[k] / calls[k] for k in t},
            "largest_error_of_a_decrypted_sum": worst, "correct": worst < 1e-9}


def main():
    import phe
    out = {"shape": "examples/federated_learning_with_encryption.py:122-133,213-225: %d encrypt + %d add + %d decrypt scalar calls"
                    % (N_CLIENTS * N_ROUNDS * N_COMPONENTS, (N_CLIENTS - 1) * N_ROUNDS * N_COMPONENTS, N_ROUNDS * N_COMPONENTS),
           "runs": [run(bits, phe) for bits in (1024, 2048)]}
    try:
        import federated_learning_batched as fed
        t0 = time.perf_counter()
        errors, elapsed = fed.run(key_length=1024, verbose=False)
        out["batched_form_same_protocol_1024_bits"] = {"seconds": time.perf_counter() - t0, "protocol_seconds": elapsed,
                                                       "errors": ["%.2f" % e for e in errors]}
    except Exception as e:  # noqa: BLE001
        out["batched_form_same_protocol_1024_bits"] = {"error": repr(e)}
    out["reference_cpu_for_scale"] = "the reference itself runs the file in 33.5 s without gmpy2 in the build container (SURVEY 0.2)"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
