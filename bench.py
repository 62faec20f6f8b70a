#!/usr/bin/env python3
"""bench.py -- homomorphic mul+relin/sec, CKKS N=2^14 (BASELINE.json metric) on N MI355X.

One "step" = one pass of the hot path (SealCiphertext::multiply = Evaluator::multiply +
relinearize_inplace, src/runtime/SealCiphertext.cpp:102-107) over a batch of B independent synthetic
ciphertext pairs resident in HBM, through the C ABI (abc_hip_mul_relin).  Multi-GPU: the independent
pairs are sharded over ranks (weak scaling, B per GPU fixed), no data-path collective; the result
ciphertexts of the last step are gathered to rank 0 over RCCL after the timed region (the one exchange
the path has).

Prints ONE JSON line (rank 0) with `roofline` (HIP-event-timed hot launch vs the 8 TB/s HBM peak,
algorithmic bytes 8N(6L+2L(L+1)) per multiply) and `cpu_baseline` (the CPU oracle port on host cores).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N = 16384
BITS = [50, 40, 40, 40, 50]  # 4 data limbs + special prime (config 3 chain)
L = 4
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
ALGO_BYTES = 8 * N * (6 * L + 2 * L * (L + 1))  # = 8 388 608 B per mul+relin


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="independent ciphertext pairs per GPU per step")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="bound of the cpu_baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    return ap.parse_args()


def cpu_baseline(primes, seconds):
    """The oracle (a port of SEAL's algorithms, kind="port") timed on this host: one mul+relin per thread at a
    time over the independent batch, on all host cores."""
    import multiprocessing as mp
    cores = os.cpu_count() or 1
    workers = min(cores, 64)
    with mp.get_context("spawn").Pool(workers) as pool:
        res = pool.starmap(_cpu_worker, [(primes, seconds, w) for w in range(workers)])
    ops = sum(r[0] for r in res)
    wall = max(r[1] for r in res)
    return {"value": ops / wall, "unit": "mul+relin/s", "cores": workers, "kind": "port",
            "sample": "%d mul+relin of the same CKKS N=16384 L=4 workload over %d threads in %.1f s" % (ops, workers, wall)}


def _cpu_worker(primes, seconds, w):
    import numpy as np
    from oracle import oracle_py as om
    o = om.Oracle(om.CKKS, N, primes)
    o.keygen(0xABC00001)
    rng = np.random.default_rng(1000 + w)
    a = np.stack([rng.integers(0, q, size=(2, N), dtype=np.uint64) for q in primes[:L]], axis=1)
    b = np.stack([rng.integers(0, q, size=(2, N), dtype=np.uint64) for q in primes[:L]], axis=1)
    o.time_mul_relin(a, b, 1)
    t0 = time.time()
    ops = 0
    while time.time() - t0 < seconds:
        o.time_mul_relin(a, b, 4)
        ops += 4
    return ops, time.time() - t0


def measured_traffic(batch):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in
    separate runs and corrected as MI355X_MICROARCH.md prescribes; newest profiles/r*_pmc_traffic.json,
    produced by tools/pmc_traffic.sh), scaled to this
    batch.  bench.py cannot collect PMC counters itself; None if the profile is absent."""
    try:
        import glob
        latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]
        with open(latest) as f:
            return json.load(f)["hbm_bytes_per_mul_relin"] * batch
    except Exception:
        return None


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import numpy as np
    import torch
    import torch.distributed as dist
    from abc_amd import capi

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    primes = capi.create_primes(N, BITS)
    g = capi.Context(capi.CKKS, N, primes, device=local_rank)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    g.keygen(0xABC00001)  # keys replicated on every device

    B = args.batch
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)

    def rand_ct():
        t = torch.empty((B, 2, L, N), dtype=torch.int64, device=dev)
        for j, q in enumerate(primes[:L]):
            t[:, :, j, :] = torch.randint(0, q, (B, 2, N), dtype=torch.int64, device=dev, generator=gen)
        return t

    a, b = rand_ct(), rand_ct()  # uniformly random residues = what ciphertexts look like
    out = torch.empty_like(a)
    pa, pb, po = (C.c_void_p(t.data_ptr()) for t in (a, b, out))

    def step():
        g.op("mul_relin", pa, pb, po, L, C.c_size_t(B))

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    g.timer_start()  # HIP events on the stream the kernels run on
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ev_ms = g.timer_stop()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed, ev_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, ev_ms = float(tt[0]), float(tt[1])
        # the path's only exchange: gather the result ciphertexts of the last step on rank 0
        from abc_amd.sharding import gather_results
        gathered = gather_results(out, B * world, dst=0)
        torch.cuda.synchronize()
        if rank == 0:
            assert gathered.shape[0] == B * world

    if rank == 0:
        total_ops = B * world * args.steps
        value = total_ops / elapsed
        launch_ms = ev_ms / args.steps  # one hot launch = one abc_hip_mul_relin over B pairs
        achieved = (ALGO_BYTES * B) / (launch_ms * 1e-3) / 1e9
        line = {
            "metric": "homomorphic mul+relin/sec, CKKS N=2^14", "value": value, "unit": "mul+relin/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64 (transforms: exact integer arithmetic in f64)", "data": "synthetic",
            "config": {"workload": "CKKS N=16384, 4 data limbs {50,40,40,40} + special 50-bit prime, ct x ct multiply + relinearize",
                       "batch_per_gpu": B, "sharding": "independent ciphertext pairs per rank, result gather only"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(B),
                         "kernel": "abc_hip_mul_relin = k_fused_tensor_pass0_fp + k_fused_tailmac_coop_fp (dominant) + "
                                   "k_fused_ks_special_intt_fp + k_fused_ks_moddown_fp",
                         "algorithmic_bytes_per_launch": ALGO_BYTES * B, "launch_ms": launch_ms},
        }
        if not args.no_cpu and world == 1:  # CPU leg on rank 0 at N=1 only
            # parity spot check of the measured path against the oracle, then the CPU leg
            from oracle import oracle_py as om
            o = om.Oracle(om.CKKS, N, primes)
            o.keygen(0xABC00001)
            ha = a[0].cpu().numpy().view(np.uint64)
            hb = b[0].cpu().numpy().view(np.uint64)
            ho = out[0].cpu().numpy().view(np.uint64)
            if not np.array_equal(o.mul_relin(ha, hb), ho):
                raise SystemExit("bench: GPU result differs from the oracle -- number is invalid")
            line["cpu_baseline"] = cpu_baseline(primes, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
