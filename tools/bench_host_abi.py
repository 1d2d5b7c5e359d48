#!/usr/bin/env python3
"""The plain host-pointer C-ABI (phe_hip_encrypt / phe_hip_decrypt / phe_hip_mulmod on numpy arrays in pageable memory)
against the device-pointer entry points on resident operands: what a maintainer who binds only the host-pointer functions
 gets.  From two chunks of 65536 rows on the library pipelines the batch through pinned staging buffers
(csrc/phe_hip.hip run_pipelined); PHE_HIP_NO_PIPELINE=1 restores the blocking copies for comparison."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "python-paillier_amd")):
    sys.path.insert(0, p)


def run(batch, key_bits):
    import numpy as np
    import torch
    from phe import _native as native
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "paillier_%d.json" % key_bits)))
    H = lambda k: int(g[k], 16)
    s1, s2 = key_bits // 32, key_bits // 16
    ctx = native.Context(H("n"), H("p"), H("q"), H("hp"), H("hq"), H("p_inverse"), n_limbs=s1)
    rng = np.random.Generator(np.random.PCG64(3))
    m = rng.integers(0, 2 ** 32, (batch, s1), dtype=np.uint64).astype(np.uint32)
    r = rng.integers(0, 2 ** 32, (batch, s1), dtype=np.uint64).astype(np.uint32)
    m[:, -1] = 0
    r[:, -1] &= 0x3fffffff
    r[:, 0] |= 1
    dev = torch.device("cuda", 0)
    md, rd = (torch.from_numpy(a.view(np.int32)).to(dev) for a in (m, r))
    cd = torch.empty((batch, s2), dtype=torch.int32, device=dev)
    bd = torch.empty((batch, s1), dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def timed(fn, reps=2):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps, out

    res = {"batch": batch, "key_bits": key_bits, "pipeline": not os.environ.get("PHE_HIP_NO_PIPELINE")}
    t_dev, _ = timed(lambda: ctx.encrypt_dev(md.data_ptr(), rd.data_ptr(), cd.data_ptr(), batch, st))
    t_host, c = timed(lambda: ctx.encrypt(m, r))
    same = bool(np.array_equal(c, cd.cpu().numpy().view(np.uint32)))
    res["encrypt"] = {"dev_per_s": batch / t_dev, "host_ptr_per_s": batch / t_host, "host_over_dev": t_dev / t_host, "same_bits": same}
    t_dev, _ = timed(lambda: ctx.decrypt_dev(cd.data_ptr(), bd.data_ptr(), batch, st))
    t_host, back = timed(lambda: ctx.decrypt(c))
    res["decrypt"] = {"dev_per_s": batch / t_dev, "host_ptr_per_s": batch / t_host, "host_over_dev": t_dev / t_host,
                      "roundtrip": bool(np.array_equal(back, m))}
    c2 = np.ascontiguousarray(c[::-1])
    t_host, prod = timed(lambda: ctx.mulmod(c, c2))
    res["mulmod"] = {"host_ptr_per_s": batch / t_host, "GBps_over_pcie": 3 * s2 * 4 * batch / t_host / 1e9}
    return res


if __name__ == "__main__":
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    bits = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    if len(sys.argv) > 3 and sys.argv[3] == "child":
        print(json.dumps(run(batch, bits)))
    else:
        out = {}
        for label, env in (("pipelined", {}), ("blocking_copies", {"PHE_HIP_NO_PIPELINE": "1"})):
            res = subprocess.run([sys.executable, os.path.abspath(__file__), str(batch), str(bits), "child"],
                                 env=dict(os.environ, **env), capture_output=True, text=True)
            line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
            out[label] = json.loads(line[-1]) if line else {"error": res.stderr[-500:]}
        print(json.dumps(out))
