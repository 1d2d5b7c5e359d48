"""Multi-GPU sharding: one process per GPU, contiguous batch shards, no data-path collective.

Every element of a Paillier batch is independent and the key constants are a few KiB, so the batch is cut into
contiguous ranges `[rank*B/G, (rank+1)*B/G)`.  Each rank runs the ordinary single-GPU entry
points on its shard.  The only exchange step that can be needed is the final concatenation of ciphertext
shards on every device: `all_gather_rows` does it with one `all_gather_into_tensor` (RCCL over xGMI when the
process group is "nccl"; the same code runs on "gloo" for the CPU tests).
"""
import numpy as np


def shard_bounds(total, world_size, rank):
    """Contiguous, balanced split: the first `total % world_size` ranks get one extra row."""
    base, extra = divmod(total, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(total, world_size):
    return [shard_bounds(total, world_size, r)[1] - shard_bounds(total, world_size, r)[0] for r in range(world_size)]


def my_shard(array, world_size, rank):
    lo, hi = shard_bounds(len(array), world_size, rank)
    return array[lo:hi]


def all_gather_rows(local_rows, total_rows, group=None):
    """Concatenate per-rank row blocks (torch tensor (rows_r, width), any integer dtype) into (total_rows, width)
    on every rank.  Shards may differ by one row; they are padded to the largest shard for the collective."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = shard_sizes(total_rows, world)
    width = local_rows.shape[1]
    biggest = max(sizes)
    padded = local_rows
    if local_rows.shape[0] < biggest:
        pad = torch.zeros((biggest - local_rows.shape[0], width), dtype=local_rows.dtype, device=local_rows.device)
        padded = torch.cat([local_rows, pad])
    out = torch.empty((world * biggest, width), dtype=local_rows.dtype, device=local_rows.device)
    dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
    if all(s == biggest for s in sizes):
        return out
    return torch.cat([out[r * biggest:r * biggest + sizes[r]] for r in range(world)])


def gather_ciphertexts(local_limbs, total_rows, group=None):
    """numpy (rows_r, ct_limbs) uint32 -> numpy (total_rows, ct_limbs) uint32 on every rank (host tensors)."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(local_limbs).view(np.int32))
    return all_gather_rows(t, total_rows, group).numpy().view(np.uint32)


def sum_over_ranks(local_sum, group=None, public_key=None):
    """The homomorphic sum of a sharded vector: every rank hands in the EncryptedNumber its own shard adds up to
    (`EncryptedVector.sum()`; None for an empty shard, then with `public_key`) and gets back the sum over all ranks — the same EncryptedNumber,
    bit for bit, on every rank.  RCCL has no reduction by modular multiplication: the G partial
    ciphertexts (G x ct_limbs words, a few KB) are all-gathered and combined locally, in rank order.  The result equals
    the reference's `sum(list_of_encrypted_numbers)` over the whole vector (phe/paillier.py:705-719 and :570-601 for the
    exponent alignment): canonical residues of the same product, whatever the order of the factors."""
    import torch.distributed as dist
    from .ciphertext import EncryptedNumber, EncryptedVector
    world = dist.get_world_size(group)
    mine = None if local_sum is None else (int(local_sum.ciphertext(be_secure=False)), int(local_sum.exponent))
    parts = [None] * world
    dist.all_gather_object(parts, mine, group=group)
    parts = [p for p in parts if p is not None]
    if not parts:
        raise ValueError("empty vector")
    pk = local_sum.public_key if local_sum is not None else public_key
    if pk is None:
        raise ValueError("a rank without a partial sum must be given the public key")
    if len(parts) == 1:
        return EncryptedNumber(pk, parts[0][0], parts[0][1])
    vec = EncryptedVector.from_ciphertexts(pk, [c for c, _ in parts], [e for _, e in parts])
    return vec.sum()


def library_communicator(ctx, rank, world_size, exchange):
    """An RCCL communicator owned by the native library (include/phe_hip.h phe_hip_comm_create) for hosts that have no
    process group of their own.  `exchange(id_bytes_or_None) -> id_bytes` is the host's channel for the 128-byte id:
    called with the id on rank 0 and with None elsewhere, it must return rank 0's bytes on every rank (a file, a
    socket, an MPI broadcast ...; with torch.distributed: dist.broadcast_object_list)."""
    from . import _native
    uid = exchange(_native.comm_unique_id() if rank == 0 else None)
    return _native.Communicator(ctx, uid, rank, world_size)


def all_gather_rows_library(comm, local_ptr, all_ptr, rows, limbs, stream=0):
    """ONE ncclAllGather issued by the library: local (rows, limbs) device words -> all (world * rows, limbs) on every
    rank, rank r at rows [r * rows, (r + 1) * rows).  Equal `rows` on every rank (pad the last shard)."""
    comm.allgather_dev(local_ptr, all_ptr, rows, limbs, stream)
