"""Sharding of independent circuits / ciphertext batches over the GPUs of one node.

The reference has no batching at all (one RuntimeVisitor = one circuit, ref:include/ast_opt/runtime/
RuntimeVisitor.h:34); independent units share only read-only evaluation keys, so they partition with no
data-path collective.  The single exchange is the gather of result ciphertexts to rank 0
(torch.distributed: RCCL over xGMI on GPUs, gloo in the CPU rehearsal tests).
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous block partition of `total` independent units: returns (start, stop) of this rank's block.
    Blocks differ by at most one unit; concatenating them in rank order restores the original order."""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def gather_results(local, total, dst=0):
    """Gather per-rank result tensors [n_local, ...] into [total, ...] on rank `dst` (None elsewhere).
    Ranks may hold different n_local (ragged tail).  Rank `dst` allocates the result once and receives every other rank's block
    straight into its slice (point-to-point send / recv: xGMI links into one GPU under RCCL, sockets under gloo) -- no padded
    staging buffers and no concatenation, so the transient footprint is the result itself (an 8-way gather of 1 GiB blocks
    through `dist.gather` + `cat` would hold 16 GiB)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(total, r, world) for r in range(world)]
    lo, hi = sizes[rank]
    if local.shape[0] != hi - lo:
        raise ValueError("rank %d holds %d units, its block of %d is %d" % (rank, local.shape[0], total, hi - lo))
    local = local.contiguous()
    if rank != dst:
        if hi > lo:
            dist.send(local, dst)
        return None
    out = torch.empty((total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    out[lo:hi] = local
    for r in range(world):
        a, b = sizes[r]
        if r != dst and b > a:
            dist.recv(out[a:b], r)  # a leading-dimension slice of a contiguous tensor is contiguous
    return out
