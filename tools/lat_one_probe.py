#!/usr/bin/env python3
"""Device time of ONE decrypt through the C-ABI (phe_hip_decrypt_dev, batch of 1) on a golden key: HIP events around 20 calls.
    python tools/lat_one_probe.py [key_bits]"""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-paillier_amd")); sys.path.insert(0, ROOT)
import torch
from phe import _native as native
ks = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
g = json.load(open(os.path.join(ROOT, "tests", "golden", "paillier_%d.json" % ks)))
H = lambda k: int(g[k], 16)
s1, s2 = ks // 32, ks // 16
ctx = native.Context(H("n"), H("p"), H("q"), H("hp"), H("hq"), H("p_inverse"), n_limbs=s1)
dev = torch.device("cuda", 0)
c = torch.randint(0, 2**31 - 1, (1, s2), dtype=torch.int32, device=dev); c[:, s2 - 1] = 0
out = torch.empty((1, s1), dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3): ctx.decrypt_dev(c.data_ptr(), out.data_ptr(), 1, st)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): ctx.decrypt_dev(c.data_ptr(), out.data_ptr(), 1, st)
b.record(); b.synchronize()
print("key", ks, "decrypt_dev batch 1: %.3f ms device time per call" % (a.elapsed_time(b) / 20), ctx.last_launch())
t0 = time.perf_counter()
for _ in range(20):
    ctx.decrypt_dev(c.data_ptr(), out.data_ptr(), 1, st); torch.cuda.synchronize()
print("  with a sync per call: %.3f ms wall" % ((time.perf_counter() - t0) / 20 * 1e3))
