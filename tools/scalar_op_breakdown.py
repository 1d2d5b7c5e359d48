#!/usr/bin/env python3
"""Where a SCALAR homomorphic operation spends its time (EncryptedNumber + EncryptedNumber, * float, one row at a time):
the operator, the engine call on Python ints, the host-pointer C-ABI call on limb arrays, the same kernels on resident rows
with one sync, and the cProfile top of the operator.   python tools/scalar_op_breakdown.py [key_bits]"""
import cProfile, io, json, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-paillier_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
from phe import paillier
from phe import _native as native

ks = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
g = json.load(open(os.path.join(ROOT, "tests", "golden", "paillier_%d.json" % ks)))
H = lambda k: int(g[k], 16)
pub = paillier.PaillierPublicKey(H("n"))
priv = paillier.PaillierPrivateKey(pub, H("p"), H("q"))
N = 500
xs = [pub.encrypt(float(i) + 0.5) for i in range(N)]
ys = [pub.encrypt(float(i) * 3.25) for i in range(N)]
def timed(label, fn, reps=N):
    fn(0); fn(1)
    t0 = time.perf_counter()
    for i in range(reps): fn(i)
    dt = (time.perf_counter() - t0) / reps * 1e6
    print("%-64s %8.1f us" % (label, dt)); return dt
out = {"key_bits": ks}
out["enc_plus_enc"] = timed("EncryptedNumber + EncryptedNumber", lambda i: xs[i] + ys[i])
out["enc_plus_float"] = timed("EncryptedNumber + float", lambda i: xs[i] + 2.5)
out["enc_times_float"] = timed("EncryptedNumber * float", lambda i: xs[i] * 3.14)
out["decrypt"] = timed("private_key.decrypt(EncryptedNumber)", lambda i: priv.decrypt(xs[i]), 200)
eng = pub._get_engine()
ca = [x.ciphertext(False) for x in xs]; cb = [y.ciphertext(False) for y in ys]
out["engine_raw_add_ints"] = timed("engine.raw_add([int], [int]) + to_ints", lambda i: eng.to_ints(eng.raw_add([ca[i]], [cb[i]])))
la = eng.cipher_limbs(ca); lb = eng.cipher_limbs(cb)
ctx = eng.ctx
out["abi_mulmod_host"] = timed("phe_hip_mulmod, host pointers, 1 row", lambda i: ctx.mulmod(la[i:i + 1], lb[i:i + 1]))
dev = torch.device("cuda", 0)
da = torch.from_numpy(la.view(np.int32)).to(dev); db = torch.from_numpy(lb.view(np.int32)).to(dev); do = torch.empty_like(da)
s2 = la.shape[1]
def resident(i):
    ctx.mulmod_dev(da.data_ptr() + i * s2 * 4, db.data_ptr() + i * s2 * 4, do.data_ptr() + i * s2 * 4, 1, 0); ctx.sync(0)
out["abi_mulmod_dev_sync"] = timed("phe_hip_mulmod_dev on resident rows + stream sync, 1 row", resident)
ha = torch.empty((1, s2), dtype=torch.int32).pin_memory()
def copies(i):
    da[0:1].copy_(ha, non_blocking=False); ha.copy_(do[0:1], non_blocking=False)
out["torch_pinned_h2d_d2h_1row"] = timed("one pinned H2D + one D2H of a row (torch)", copies)
pr = cProfile.Profile(); pr.enable()
for i in range(N): xs[i] + ys[i]
pr.disable(); s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14); print(s.getvalue()[:3000])
print(json.dumps(out))
