#!/usr/bin/env python3
"""Bulk serialisation of a ciphertext vector (docs/serialisation.rst:24-43 of the reference):
the decimal conversion kernels (kernels_radix.hip) and the end-to-end EncryptedVector.to_json / from_json, next to the
reference's way of producing the same text (str(int) per ciphertext + json.dumps) timed on a sample of the same vector.
Usage (GPU box): python tools/bench_wire.py [--batch 262144] [--key-bits 2048].  Prints one JSON object."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "python-paillier_amd")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1 << 18)
    ap.add_argument("--key-bits", type=int, default=2048)
    ap.add_argument("--cpu-sample", type=int, default=20000)
    args = ap.parse_args()
    import numpy as np
    import torch
    from phe import _native as native
    from phe import paillier
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "paillier_%d.json" % args.key_bits)))
    H = lambda k: int(g[k], 16)
    pub = paillier.PaillierPublicKey(H("n"))
    eng = pub._get_engine()
    ctx = eng.ctx
    B, words = args.batch, eng.ct_limbs
    width = ctx.decimal_width(words)
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev); gen.manual_seed(11)
    m = torch.randint(-2 ** 31, 2 ** 31, (B, eng.n_limbs), dtype=torch.int32, device=dev, generator=gen)
    r = torch.randint(-2 ** 31, 2 ** 31, (B, eng.n_limbs), dtype=torch.int32, device=dev, generator=gen)
    m[:, -1] = 0; r[:, -1] &= 0x3fffffff; r[:, 0] |= 1
    c = torch.empty((B, words), dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    ctx.encrypt_dev(m.data_ptr(), r.data_ptr(), c.data_ptr(), B, st)
    torch.cuda.synchronize()
    digits = torch.empty((B, width), dtype=torch.uint8, device=dev)
    back = torch.empty_like(c)
    L = native.lib()

    def kernel_time(fn, reps=3):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    import ctypes
    bad = ctypes.c_size_t(0)
    t_to = kernel_time(lambda: native._check(L.phe_hip_to_decimal_dev(ctx._h, c.data_ptr(), words, digits.data_ptr(), width, B, st)))
    t_from = kernel_time(lambda: native._check(L.phe_hip_from_decimal_dev(ctx._h, digits.data_ptr(), width, back.data_ptr(), words, B,
                                                                         ctypes.byref(bad), st)))
    res = {"key_bits": args.key_bits, "batch": B, "ciphertext_words": words, "digits_per_ciphertext": width,
           "k_to_decimal": {"numbers_per_s": B / t_to, "ms": t_to * 1e3, "hbm_GBps_algorithmic": B * (4 * words + width) / t_to / 1e9},
           "k_from_decimal": {"numbers_per_s": B / t_from, "ms": t_from * 1e3, "hbm_GBps_algorithmic": B * (4 * words + width) / t_from / 1e9},
           "roundtrip_bit_exact": bool(torch.equal(back, c))}
    # the digits against Python's on a strided sample
    idx = torch.arange(0, B, max(1, B // 256), device=dev)[:256]
    ints = native.limbs_to_ints(c[idx].cpu().numpy().view(np.uint32))
    got = [bytes(row).decode().lstrip("0") or "0" for row in digits[idx].cpu().numpy()]
    res["digits_equal_python_str_sample"] = got == [str(v) for v in ints]
    # ---- end to end: EncryptedVector.to_json / from_json on the resident vector ----
    from phe._device import DeviceArray
    darr = DeviceArray(ctx, B, words, _ptr=c.data_ptr())
    vec = paillier.EncryptedVector(pub, darr, np.zeros(B, dtype=np.int64), obfuscated=True)
    t0 = time.perf_counter(); text = vec.to_json(); t_json = time.perf_counter() - t0
    t0 = time.perf_counter(); vec2 = paillier.EncryptedVector.from_json(text, device=True); t_load = time.perf_counter() - t0
    same = bool(np.array_equal(vec2._limbs.to_host(), c.cpu().numpy().view(np.uint32)))
    res["to_json"] = {"ciphertexts_per_s": B / t_json, "seconds": t_json, "text_MB": len(text) / 1e6}
    res["from_json"] = {"ciphertexts_per_s": B / t_load, "seconds": t_load, "same_limbs": same}
    # ---- the reference's way on one core: ints -> str per element + json.dumps; json.loads + int(str) ----
    k = min(B, args.cpu_sample)
    sample = native.limbs_to_ints(c[:k].cpu().numpy().view(np.uint32))
    t0 = time.perf_counter()
    ref_text = json.dumps({"public_key": {"n": pub.n}, "values": [[str(v), 0] for v in sample]})
    t_ref = time.perf_counter() - t0
    t0 = time.perf_counter()
    vals = [int(v[0]) for v in json.loads(ref_text)["values"]]
    t_ref_load = time.perf_counter() - t0
    res["cpu_reference_way"] = {"sample": k, "to_json_ciphertexts_per_s": k / t_ref, "from_json_ciphertexts_per_s": k / t_ref_load,
                                "roundtrip": vals == sample,
                                "same_text_prefix": text[:len(ref_text) - 2] == ref_text[:-2]}
    res["speedup_to_json"] = res["to_json"]["ciphertexts_per_s"] / res["cpu_reference_way"]["to_json_ciphertexts_per_s"]
    res["speedup_from_json"] = res["from_json"]["ciphertexts_per_s"] / res["cpu_reference_way"]["from_json_ciphertexts_per_s"]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
