"""N>1 path on CPU: two gloo ranks each process their contiguous shard (emulator backend standing in for
the GPU) and all-gather the ciphertext shards; the concatenation must equal the oracle on the whole batch."""
import os
import subprocess
import sys
import textwrap

import numpy as np

from conftest import PKG, ROOT

if PKG not in sys.path:
    sys.path.insert(0, PKG)

from phe import sharding  # noqa: E402


def test_shard_bounds_cover_everything():
    for total in (0, 1, 7, 8, 9, 1000003):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == sharding.shard_sizes(total, world)


WORKER = textwrap.dedent('''
    import os, sys, json
    import numpy as np
    sys.path[:0] = [os.environ["PHE_ROOT"], os.path.join(os.environ["PHE_ROOT"], "tests"),
                    os.path.join(os.environ["PHE_ROOT"], "python-paillier_amd")]
    import torch.distributed as dist
    import emu_backend
    emu_backend.install()
    from phe import paillier, sharding, _native
    from oracle.paillier_oracle import COracle
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    g = json.load(open(os.path.join(os.environ["PHE_ROOT"], "tests", "golden", "paillier_256.json")))
    n = int(g["n"], 16)
    pub = paillier.PaillierPublicKey(n)
    rng = np.random.Generator(np.random.PCG64(99))            # same stream on every rank
    total = 7                                                  # uneven split on purpose
    m = [int(x) for x in rng.integers(0, 2 ** 62, total)]
    r = [int(x) + 1 for x in rng.integers(0, 2 ** 62, total)]
    lo, hi = sharding.shard_bounds(total, world, rank)
    eng = pub._get_engine()
    local = eng.raw_encrypt(m[lo:hi], r[lo:hi])
    full = sharding.gather_ciphertexts(local, total)
    want = COracle().encrypt(_native.int_to_limbs(n, 8), _native.ints_to_limbs(m, 8), _native.ints_to_limbs(r, 8))
    assert full.shape == want.shape and np.array_equal(full, want), "rank %d mismatch" % rank
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
''')


def _run_ranks(tmp_path, world, port):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, PHE_ROOT=ROOT, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert res.stdout.count("ok") == world


def test_two_rank_gloo_encrypt_and_allgather(tmp_path):
    _run_ranks(tmp_path, 2, 29617)


def test_three_rank_gloo_encrypt_and_allgather(tmp_path):
    """7 rows over 3 ranks (3 + 2 + 2): ragged shards, padding rows in the gather, an odd world size — the first hardware run
    with N > 2 must not be the first time this cut is made"""
    _run_ranks(tmp_path, 3, 29619)


SUM_WORKER = textwrap.dedent('''
    import os, sys, json
    import numpy as np
    sys.path[:0] = [os.environ["PHE_ROOT"], os.path.join(os.environ["PHE_ROOT"], "tests"),
                    os.path.join(os.environ["PHE_ROOT"], "python-paillier_amd")]
    import torch.distributed as dist
    import emu_backend
    emu_backend.install()
    from phe import paillier, sharding
    from phe.ciphertext import EncryptedVector
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    g = json.load(open(os.path.join(os.environ["PHE_ROOT"], "tests", "golden", "paillier_256.json")))
    n, p, q = (int(g[k], 16) for k in ("n", "p", "q"))
    pub = paillier.PaillierPublicKey(n)
    priv = paillier.PaillierPrivateKey(pub, p, q)
    rng = np.random.Generator(np.random.PCG64(7))              # same stream on every rank
    total = 7
    values = [float(x) for x in rng.normal(size=total)] [:4] + [int(x) for x in rng.integers(-1000, 1000, total - 4)]   # mixed exponents
    r = [int(x) + 1 for x in rng.integers(0, 2 ** 62, total)]
    lo, hi = sharding.shard_bounds(total, world, rank)
    shard = pub.encrypt_batch(values[lo:hi], r_values=r[lo:hi])
    got = sharding.sum_over_ranks(shard.sum())
    # the reference's chain over the WHOLE vector on one rank: sum() of EncryptedNumbers, left to right
    whole = pub.encrypt_batch(values, r_values=r).to_numbers()
    want = whole[0]
    for x in whole[1:]:
        want = want + x
    assert got.exponent == want.exponent and got.ciphertext(False) == want.ciphertext(False), "rank %d: bits differ" % rank
    assert abs(priv.decrypt(got) - sum(values)) < 1e-9
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
''')


def test_homomorphic_sum_over_the_ranks_equals_the_reference_chain(tmp_path):
    """a sum over a sharded vector is G partial products all-gathered and combined locally (RCCL has no
    reduction by modular multiplication) — the same bits on every rank as the reference's left-to-right chain over the whole
    vector, mixed exponents included; three ranks, ragged shards (3 + 2 + 2)"""
    script = tmp_path / "sum_worker.py"
    script.write_text(SUM_WORKER)
    env = dict(os.environ, PHE_ROOT=ROOT, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3",
           "--master-addr", "127.0.0.1", "--master-port", "29623", str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert res.stdout.count("ok") == 3
