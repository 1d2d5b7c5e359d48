#!/usr/bin/env python3
"""BASELINE.json configs[3]: 3072-bit key, 8M-plaintext encrypt sharded across the GPUs of one node, the ciphertext
shards concatenated on every GPU by ONE RCCL all-gather over xGMI (the only exchange step of the path).

One process per GPU:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29511 \
        tools/bench_config4.py --total 8388608
(N = 1 works too and is what a 1-GPU box can show: --total 1048576 is one GPU's share of the 8-GPU job.)
Each rank encrypts the contiguous shard [rank*B/N, (rank+1)*B/N) of a seeded batch (operands generated on the device),
then `phe.sharding.all_gather_rows` builds the (B, 192)-word vector on every rank.  Rank 0 compares the rows at every
shard boundary and a strided sample with the libgmp oracle and prints one JSON object."""
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
    ap.add_argument("--total", type=int, default=1 << 23, help="plaintexts in the whole job (all ranks together)")
    ap.add_argument("--key-bits", type=int, default=3072, choices=[1024, 2048, 3072])
    args = ap.parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist
    from phe import _native as native
    from phe.sharding import all_gather_rows, shard_bounds

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    g = json.load(open(os.path.join(ROOT, "tests", "golden", "paillier_%d.json" % args.key_bits)))
    n_int = int(g["n"], 16)
    s1, s2 = args.key_bits // 32, args.key_bits // 16
    ctx = native.Context(n_int, device=local_rank, n_limbs=s1)
    B = args.total
    lo, hi = shard_bounds(B, world, rank)
    rows = hi - lo

    def operands(first, count):
        """rows [first, first+count) of the job's (m, r): a function of the row index only, so any rank (and the
        checker) can regenerate any row — a counter-based stream seeded per 2^16-row block"""
        ms, rs = [], []
        blk = 1 << 16
        for b in range(first // blk, (first + count - 1) // blk + 1):
            gen = torch.Generator(device=dev)
            gen.manual_seed(4000 + b)
            mb = torch.randint(-2 ** 31, 2 ** 31, (blk, s1), dtype=torch.int32, device=dev, generator=gen)
            rb = torch.randint(-2 ** 31, 2 ** 31, (blk, s1), dtype=torch.int32, device=dev, generator=gen)
            a, z = max(first, b * blk) - b * blk, min(first + count, (b + 1) * blk) - b * blk
            ms.append(mb[a:z]); rs.append(rb[a:z])
        m, r = torch.cat(ms), torch.cat(rs)
        m[:, s1 - 1] = 0
        r[:, s1 - 1] &= 0x3fffffff
        r[:, 0] |= 1
        return m.contiguous(), r.contiguous()

    m, r = operands(lo, rows)
    c = torch.empty((rows, s2), dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def sync():
        torch.cuda.synchronize()
        dist.barrier()

    sync()
    t0 = time.perf_counter()
    ctx.encrypt_dev(m.data_ptr(), r.data_ptr(), c.data_ptr(), rows, st)
    sync()
    t_enc = time.perf_counter() - t0
    t0 = time.perf_counter()
    full = all_gather_rows(c, B)
    sync()
    t_gather = time.perf_counter() - t0
    times = torch.tensor([t_enc, t_gather], dtype=torch.float64, device=dev)
    dist.all_reduce(times, op=dist.ReduceOp.MAX)
    t_enc, t_gather = (float(x) for x in times.tolist())

    ok = None
    if rank == 0:
        from oracle.paillier_oracle import COracle
        orc = COracle()
        idx = sorted(set([0, B - 1] + [shard_bounds(B, world, k)[0] for k in range(world)] +
                         [max(0, shard_bounds(B, world, k)[1] - 1) for k in range(world)] +
                         list(range(0, B, max(1, B // 24)))))
        to_np = lambda t: t.cpu().numpy().view(np.uint32)
        ms, rs = zip(*[operands(i, 1) for i in idx])
        want = orc.encrypt(native.int_to_limbs(n_int, s1), to_np(torch.cat(ms)), to_np(torch.cat(rs)), nthreads=8)
        ok = bool(np.array_equal(to_np(full[torch.tensor(idx, device=dev)]), want))
        print(json.dumps({
            "config": "configs[3]: %d-bit key, %d plaintexts sharded over %d GPU(s), one RCCL all-gather of the ciphertext shards"
                      % (args.key_bits, B, world),
            "n_gpus": world, "total": B, "rows_per_gpu": rows,
            "encrypt": {"seconds": t_enc, "encrypts_per_s_all_gpus": B / t_enc, "encrypts_per_s_per_gpu": B / t_enc / world},
            "all_gather": {"seconds": t_gather, "bytes_received_per_gpu": B * s2 * 4,
                           "GBps_per_gpu": B * s2 * 4 / t_gather / 1e9},
            "end_to_end_encrypts_per_s": B / (t_enc + t_gather),
            "bit_exact_boundaries_and_sample": ok, "rows_checked": len(idx), "geometry": ctx.info()}))
    dist.barrier()
    dist.destroy_process_group()
    if ok is False:
        sys.exit(1)


if __name__ == "__main__":
    main()
