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
    Ranks may hold different n_local (ragged tail); shards are padded to the largest block for the collective."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(total, r, world) for r in range(world)]
    biggest = max(b - a for a, b in sizes)
    padded = local
    if local.shape[0] < biggest:
        pad = torch.zeros((biggest - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        padded = torch.cat([local, pad], dim=0)
    bufs = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded.contiguous(), bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([bufs[r][: sizes[r][1] - sizes[r][0]] for r in range(world)], dim=0)
