#!/usr/bin/env python3
"""Config 3 of BASELINE.json: 2048-bit key, 2^20 resident ciphertexts, homomorphic add (_raw_add = mulmod mod n^2)
and scalar multiplication (_raw_mul: float-like 56-bit scalars, int64 scalars, and a 10 % negative mix that
takes the inverse branch of phe/paillier.py:745-749), and the encrypted dot product (phe_hip_multiexp) against the
powmod + product-tree composition it replaces.
Prints one JSON object; every result is checked against the libgmp oracle on a strided sample."""
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
    ap.add_argument("--batch", type=int, default=1 << 20)
    ap.add_argument("--key-bits", type=int, default=2048)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--chunk-sweep", action="store_true", help="time the dot product at forced chunk sizes (PHE_HIP_MULTI_CHUNK)")
    args = ap.parse_args()
    import numpy as np
    import torch
    from phe import _native as native
    from oracle.paillier_oracle import COracle
    dev = torch.device("cuda", 0)
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "paillier_%d.json" % args.key_bits)))
    H = lambda k: int(g[k], 16)
    n_int = H("n")
    s1, s2 = args.key_bits // 32, args.key_bits // 16
    ctx = native.Context(n_int, H("p"), H("q"), H("hp"), H("hq"), H("p_inverse"), n_limbs=s1)
    B = args.batch
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    rnd = lambda cols: torch.randint(-2 ** 31, 2 ** 31, (B, cols), dtype=torch.int32, device=dev, generator=gen)
    m, r = rnd(s1), rnd(s1)
    m[:, s1 - 1] = 0; r[:, s1 - 1] &= 0x3fffffff; r[:, 0] |= 1
    ca = torch.empty((B, s2), dtype=torch.int32, device=dev)
    cb = torch.empty_like(ca); out = torch.empty_like(ca)
    st = torch.cuda.current_stream().cuda_stream
    ctx.encrypt_dev(m.data_ptr(), r.data_ptr(), ca.data_ptr(), B, st)
    ctx.encrypt_dev(r.data_ptr(), m.data_ptr(), cb.data_ptr(), B, st)   # a second, different batch (m, r swapped roles)
    torch.cuda.synchronize()
    orc = COracle()
    n_arr = native.int_to_limbs(n_int, s1)
    to_np = lambda t: t.cpu().numpy().view(np.uint32)
    idx = torch.arange(0, B, max(1, B // 48), device=dev)[:48]

    def timed(fn):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.reps

    res = {"key_bits": args.key_bits, "batch": B, "geometry": ctx.info()}
    # ---- _raw_add ----
    t = timed(lambda: ctx.mulmod_dev(ca.data_ptr(), cb.data_ptr(), out.data_ptr(), B, st))
    ok = np.array_equal(to_np(out[idx]), orc.add(n_arr, to_np(ca[idx]), to_np(cb[idx]), nthreads=8))
    res["raw_add"] = {"ops_per_s": B / t, "ms": t * 1e3, "bit_exact_sample": bool(ok),
                      "hbm_GBps_algorithmic": 3 * s2 * 4 * B / t / 1e9}
    # ---- obfuscate (phe/paillier.py:603-624): c * r^n over an existing ciphertext ----
    t = timed(lambda: ctx.obfuscate_dev(ca.data_ptr(), r.data_ptr(), out.data_ptr(), B, st))
    ok = np.array_equal(to_np(out[idx]), orc.obfuscate(n_arr, to_np(ca[idx]), to_np(r[idx]), nthreads=8))
    res["obfuscate"] = {"ops_per_s": B / t, "ms": t * 1e3, "bit_exact_sample": bool(ok)}
    # ---- add a plaintext (phe/paillier.py:673-675): c * (1 + n*m) ----
    t = timed(lambda: ctx.add_plain_dev(ca.data_ptr(), m.data_ptr(), out.data_ptr(), B, st))
    ones = np.zeros((len(idx), s1), np.uint32); ones[:, 0] = 1
    ok = np.array_equal(to_np(out[idx]), orc.add(n_arr, to_np(ca[idx]), orc.encrypt(n_arr, to_np(m[idx]), ones, nthreads=8), nthreads=8))
    res["add_plain"] = {"ops_per_s": B / t, "ms": t * 1e3, "bit_exact_sample": bool(ok)}
    # ---- _raw_mul, positive scalars ----
    for name, bits in (("raw_mul_float56", 56), ("raw_mul_int64", 63)):
        e = rnd(2)
        if bits == 56:
            e[:, 1] &= 0x00ffffff
        else:
            e[:, 1] &= 0x7fffffff
        t = timed(lambda: ctx.powmod_dev(ca.data_ptr(), e.data_ptr(), 2, bits, out.data_ptr(), B, st))
        sc = np.zeros((len(idx), s1), np.uint32); sc[:, :2] = to_np(e[idx])
        ok = np.array_equal(to_np(out[idx]), orc.mul(n_arr, to_np(ca[idx]), sc, nthreads=8))
        res[name] = {"ops_per_s": B / t, "ms": t * 1e3, "bit_exact_sample": bool(ok)}
    # ---- _raw_mul with 10 % negative scalars: inverse of that subset, then powmod with n - s ----
    neg_rows = torch.arange(0, B, 10, device=dev)
    sub = ca[neg_rows].contiguous()
    t0 = time.perf_counter()
    inv_host = ctx.invert(to_np(sub))                      # host-pointer entry (tree of mulmod launches + 1 host inverse)
    t_inv = time.perf_counter() - t0
    ok = np.array_equal(ctx.mulmod(inv_host[:64], to_np(sub[:64]))[:, 0], np.ones(64, np.uint32))
    res["invert_10pct_subset"] = {"rows": int(len(neg_rows)), "seconds_incl_pcie": t_inv, "rows_per_s": len(neg_rows) / t_inv,
                                  "a_times_inverse_is_one": bool(ok)}
    # ---- encrypted dot product sum_i k_i * E(x_i) (np.dot over ciphertexts): one multi-exponentiation against the
    #      composition it replaces (powmod per element + log2(B) pairwise-product launches); same bits ----
    one_row = torch.empty((1, s2), dtype=torch.int32, device=dev)
    for name, bits in (("dot_float56", 56), ("dot_int64", 63)):
        e = rnd(2)
        e[:, 1] &= (0x00ffffff if bits == 56 else 0x7fffffff)
        t = timed(lambda: ctx.multiexp_dev(ca.data_ptr(), e.data_ptr(), 2, bits, one_row.data_ptr(), B, st))

        def composed():
            ctx.powmod_dev(ca.data_ptr(), e.data_ptr(), 2, bits, out.data_ptr(), B, st)
            cur = B
            while cur > 1:                                   # B is a power of two here
                half = cur // 2
                ctx.mulmod_dev(out.data_ptr(), out.data_ptr() + half * s2 * 4, out.data_ptr(), half, st)
                cur = half
        t_old = timed(composed)
        same = bool(torch.equal(out[0], one_row[0]))
        k = 256                                              # prefix against the libgmp oracle
        ctx.multiexp_dev(ca.data_ptr(), e.data_ptr(), 2, bits, one_row.data_ptr(), k, st)
        torch.cuda.synchronize()
        sc = np.zeros((k, s1), np.uint32); sc[:, :2] = to_np(e[:k])
        terms = native.limbs_to_ints(orc.mul(n_arr, to_np(ca[:k]), sc, nthreads=8))
        want = 1
        for v in terms:
            want = want * v % (n_int * n_int)
        ok = native.limbs_to_ints(to_np(one_row))[0] == want
        res[name] = {"elements_per_s": B / t, "ms": t * 1e3, "composed_powmod_plus_tree_ms": t_old * 1e3,
                     "speedup_vs_composed": t_old / t, "same_bits_as_composed": same, "bit_exact_prefix_vs_oracle": bool(ok)}
        # the same dot product over the vector resident in the pair form (phe_hip_pair_multiexp_rows_dev: no conversion in)
        pw = ctx.pair_words()
        if pw:
            pa = torch.empty((B, pw), dtype=torch.int32, device=dev)
            ctx.to_pair_dev(ca.data_ptr(), pa.data_ptr(), B, st)
            ctx.multiexp_dev(ca.data_ptr(), e.data_ptr(), 2, bits, one_row.data_ptr(), B, st)
            pair_row = torch.empty_like(one_row)
            tp = timed(lambda: ctx.pair_multiexp_rows_dev(pa.data_ptr(), e.data_ptr(), 2, bits, pair_row.data_ptr(), B, 1, st))
            res[name + "_pair_in"] = {"elements_per_s": B / tp, "ms": tp * 1e3, "speedup_vs_plain_input": t / tp,
                                      "same_bits_as_plain_input": bool(torch.equal(pair_row, one_row))}
            del pa
    # ---- matrix form: a (rows x features) plaintext matrix times an encrypted feature vector (the shape of
    #      examples/logistic_regression_encrypted_model.py:170-177 with the roles of weights and samples swapped) ----
    feat, nrows = 128, 8192
    em = torch.randint(-2 ** 31, 2 ** 31, (nrows * feat, 2), dtype=torch.int32, device=dev, generator=gen)
    em[:, 1] &= 0x00ffffff
    outm = torch.empty((nrows, s2), dtype=torch.int32, device=dev)
    t = timed(lambda: ctx.multiexp_rows_dev(ca.data_ptr(), None, em.data_ptr(), None, 2, 56, outm.data_ptr(), feat, nrows, st))

    def looped(k):
        for r in range(k):
            ctx.multiexp_dev(ca.data_ptr(), em.data_ptr() + r * feat * 8, 2, 56, out.data_ptr() + r * s2 * 4, feat, st)
    t_loop = timed(lambda: looped(256)) * (nrows / 256)
    same = bool(torch.equal(outm[:256], out[:256]))
    res["matvec_128x8192_float56"] = {"entries_per_s": feat * nrows / t, "rows_per_s": nrows / t, "ms": t * 1e3,
                                      "row_by_row_dot_ms_extrapolated_from_256_rows": t_loop * 1e3, "speedup_vs_row_by_row": t_loop / t,
                                      "same_bits_as_row_by_row": same}
    # the same matrix on one table set for the whole vector (phe_hip_multiexp_csr_dev, dense rows), and a SPARSE matrix:
    # 8192 samples x 4096 features, 100 stored entries per row (bag-of-words shaped), CSR
    t = timed(lambda: ctx.multiexp_csr_dev(ca.data_ptr(), None, feat, None, None, em.data_ptr(), None, 2, 56, None,
                                           outm.data_ptr(), nrows, st))
    res["matvec_128x8192_float56_table_form"] = {"entries_per_s": feat * nrows / t, "rows_per_s": nrows / t, "ms": t * 1e3,
                                                  "same_bits_as_chunked_form": bool(torch.equal(outm[:256], out[:256]))}
    ncols, per_row = 4096, 100
    row_ptr = torch.arange(0, (nrows + 1) * per_row, per_row, dtype=torch.int64, device=dev)
    cols = torch.randint(0, ncols, (nrows * per_row,), dtype=torch.int32, device=dev, generator=gen)
    es = torch.randint(-2 ** 31, 2 ** 31, (nrows * per_row, 2), dtype=torch.int32, device=dev, generator=gen)
    es[:, 1] &= 0x00ffffff
    outs = torch.empty((nrows, s2), dtype=torch.int32, device=dev)
    t = timed(lambda: ctx.multiexp_csr_dev(ca.data_ptr(), None, ncols, row_ptr.data_ptr(), cols.data_ptr(), es.data_ptr(), None,
                                           2, 56, None, outs.data_ptr(), nrows, st))
    # row 0 against the single-row entry point on the gathered ciphertexts
    idx0 = cols[:per_row].long()
    gathered = ca[idx0].contiguous()
    ctx.multiexp_dev(gathered.data_ptr(), es.data_ptr(), 2, 56, one_row.data_ptr(), per_row, st)
    torch.cuda.synchronize()
    res["matvec_sparse_8192x4096_100_per_row"] = {"entries_per_s": nrows * per_row / t, "rows_per_s": nrows / t, "ms": t * 1e3,
                                                   "row0_same_bits_as_single_row_entry": bool(torch.equal(outs[0], one_row[0]))}
    if args.chunk_sweep:
        e = rnd(2)
        e[:, 1] &= 0x7fffffff
        sweep = {}
        for chunk in (1, 2, 4, 8, 16, 32):
            os.environ["PHE_HIP_MULTI_CHUNK"] = str(chunk)
            t = timed(lambda: ctx.multiexp_dev(ca.data_ptr(), e.data_ptr(), 2, 63, one_row.data_ptr(), B, st))
            sweep[str(chunk)] = {"ms": t * 1e3, "elements_per_s": B / t}
        os.environ.pop("PHE_HIP_MULTI_CHUNK", None)
        res["dot_int64_chunk_sweep"] = sweep
    print(json.dumps(res))


if __name__ == "__main__":
    main()
