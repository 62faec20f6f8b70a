"""N>1 path rehearsed on CPU: world_size-2 gloo processes shard independent ciphertext batches and gather the
results on rank 0 exactly as bench.py does over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from abc_amd.sharding import gather_results, shard_range


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 256, 1000):
        for world in (1, 2, 3, 4, 8):
            blocks = [shard_range(total, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # the "circuit": each unit's result is a deterministic function of its global index, computed by the
        # CPU oracle's modular add so the shards carry real residues
        from oracle import oracle_py as om
        n = 1024
        primes = om.create_primes(n, [40, 41])
        o = om.Oracle(om.CKKS, n, primes)
        a, b = shard_range(total, rank, world)
        rng_units = []
        for g in range(a, b):
            rng = np.random.default_rng(g)
            x = rng.integers(0, primes[0], size=(2, 1, n), dtype=np.uint64)
            rng_units.append(o.add(x, x))
        local = torch.from_numpy(np.stack(rng_units).view(np.int64)) if rng_units else torch.zeros((0, 2, 1, n), dtype=torch.int64)
        res = gather_results(local, total, dst=0)
        if rank == 0:
            ok = res.shape[0] == total
            for g in range(total):
                rng = np.random.default_rng(g)
                x = rng.integers(0, primes[0], size=(2, 1, n), dtype=np.uint64)
                ok = ok and np.array_equal(res[g].numpy().view(np.uint64), o.add(x, x))
            q.put(bool(ok))
        else:
            assert res is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [6, 5])
def test_two_rank_shard_and_gather(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=10) is True
