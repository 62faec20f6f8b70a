#!/usr/bin/env python3
"""bench.py -- homomorphic mul+relin/sec, CKKS N=2^14 (BASELINE.json metric) on N MI355X.

One "step" = one pass of the hot path (SealCiphertext::multiply = Evaluator::multiply +
relinearize_inplace, src/runtime/SealCiphertext.cpp:102-107) over this rank's share of a batch of
independent synthetic ciphertext pairs resident in HBM, through the C ABI (abc_hip_mul_relin).

Multi-GPU: the independent pairs are sharded over ranks, one process per GPU, no
data-path collective; the result ciphertexts of the last step are gathered to rank 0 over RCCL after the
timed region and timed separately (`gather_ms`).  Default = STRONG scaling: `--total-batch` pairs (8192)
are split over the ranks, so per-GPU work shrinks as N grows; `--batch B` instead fixes B pairs per rank
(weak scaling).

Launch: `python bench.py --gpus N` starts the N ranks itself (fresh child processes, spawned before this
process touches torch or HIP); under `torch.distributed.run` (RANK set) it runs as the rank it is told.

Prints ONE JSON line (rank 0) with `roofline` (HIP-event-timed hot launch vs the 8 TB/s HBM peak,
algorithmic bytes 8N(6L+2L(L+1)) per multiply) and, at N=1, `cpu_baseline` (the CPU oracle port on host
cores: one thread and all cores).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N = 16384
BITS = [50, 40, 40, 40, 50]  # 4 data limbs + special prime (config 3 chain)
BITS_60 = [60, 40, 40, 40, 60]  # a SEAL-typical chain: 60-bit primes take the integer kernels, not the fp64 ones
L = 4
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
ALGO_BYTES = 8 * N * (6 * L + 2 * L * (L + 1))  # = 8 388 608 B per mul+relin
ALGO_BYTES_KEY_RESIDENT = 8 * N * 6 * L  # = 3 145 728 B: the same with the relin key counted as cache-resident across the batch


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--total-batch", type=int, default=8192,
                    help="strong scaling (default): independent ciphertext pairs per step over ALL ranks")
    ap.add_argument("--batch", type=int, default=0, help="weak scaling: pairs per GPU per step (overrides --total-batch)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="bound of each cpu_baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-60bit", action="store_true", help="skip the 60-bit-prime (integer kernel) side measurement")
    ap.add_argument("--no-bfv", action="store_true", help="skip the BFV default-ring side measurement")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="rehearse launch + sharding + gather with gloo on the CPU (no device work, no timing claims)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------
# launcher: N fresh rank processes, started before anything in this process initialises the GPU
# ---------------------------------------------------------------------------------------------------------------------
def launch_ranks(args):
    from abc_amd.launcher import launch_ranks as _launch  # imports neither torch nor HIP
    return _launch(__file__, sys.argv[1:], args.gpus)


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle: a port of SEAL 3.6's algorithms; no SEAL binary or source exists in this image)
# ---------------------------------------------------------------------------------------------------------------------
def _cpu_worker(primes, seconds, w):
    import numpy as np
    from oracle import oracle_py as om
    o = om.Oracle(om.CKKS, N, primes)
    o.keygen(0xABC00001)
    rng = np.random.default_rng(1000 + w)
    a = np.stack([rng.integers(0, q, size=(2, N), dtype=np.uint64) for q in primes[:L]], axis=1)
    b = np.stack([rng.integers(0, q, size=(2, N), dtype=np.uint64) for q in primes[:L]], axis=1)
    o.time_mul_relin(a, b, 1)
    per_op = []  # seconds per mul+relin, one sample per group of 4 (scratch preallocated outside the clock)
    t0 = time.time()
    while time.time() - t0 < seconds:
        per_op.append(o.time_mul_relin(a, b, 4) / 4)
    return per_op, time.time() - t0


def cpu_baseline(primes, seconds):
    """One mul+relin per thread at a time over the independent batch: first ONE thread alone (the figure that
    compares with a single-threaded seal::Evaluator), then all host cores.  `value` = the all-core aggregate."""
    import multiprocessing as mp
    import statistics
    single, _ = _cpu_worker(primes, min(seconds, 5.0), 0)
    cores = os.cpu_count() or 1
    workers = min(cores, 64)
    with mp.get_context("spawn").Pool(workers) as pool:
        res = pool.starmap(_cpu_worker, [(primes, seconds, w) for w in range(workers)])
    ops = 4 * sum(len(r[0]) for r in res)
    wall = max(r[1] for r in res)
    med_all = statistics.median([t for r in res for t in r[0]])
    return {"value": ops / wall, "unit": "mul+relin/s", "cores": workers, "kind": "port",
            "one_thread": {"value": 1.0 / statistics.median(single), "median_ms_per_op": statistics.median(single) * 1e3,
                           "runs": 4 * len(single)},
            "median_ms_per_op_all_cores_busy": med_all * 1e3,
            "seal_binary": "absent: no SEAL build or source in this image, the port restates SEAL 3.6's algorithms",
            "sample": "%d mul+relin of the same CKKS N=16384 L=4 workload over %d threads in %.1f s "
                      "(plus %d on one thread alone)" % (ops, workers, wall, 4 * len(single))}


def measured_traffic(batch):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in
    separate runs and corrected as MI355X_MICROARCH.md prescribes; newest profiles/r*_pmc_traffic.json,
    produced by tools/pmc_traffic.sh), scaled to this batch.  bench.py cannot collect PMC counters itself: the figure is
    NOT measured in this run, and `source` says which profile (and which batch) it was scaled from.  (None, None) if absent."""
    try:
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
            with open(path) as f:
                prof = json.load(f)
            if "hbm_bytes_per_mul_relin" in prof:  # profiles of other operations carry other keys
                src = "%s: %.2f MB per mul+relin measured at batch %s under rocprofv3 --pmc, scaled by this run's batch" % (
                    os.path.relpath(path, ROOT), prof["hbm_bytes_per_mul_relin"] / 1e6, prof.get("hot_call", {}).get("batch", "?"))
                return prof["hbm_bytes_per_mul_relin"] * batch, src
        return None, None
    except Exception:
        return None, None


def my_share(args, rank, world):
    """(first pair, number of pairs) of this rank, and the scaling mode"""
    from abc_amd.sharding import shard_range
    if args.batch > 0:
        return rank * args.batch, args.batch, args.batch * world, "weak"
    lo, hi = shard_range(args.total_batch, rank, world)
    return lo, hi - lo, args.total_batch, "strong"


# ---------------------------------------------------------------------------------------------------------------------
# CPU rehearsal of the multi-rank plumbing (tests/test_bench_launcher.py): gloo, no device, no numbers claimed
# ---------------------------------------------------------------------------------------------------------------------
def dry_run_cpu(args, rank, world):
    import torch
    import torch.distributed as dist
    from abc_amd.sharding import gather_results
    if world > 1:
        dist.init_process_group("gloo")
    lo, B, total, mode = my_share(args, rank, world)
    local = torch.arange(lo, lo + B, dtype=torch.int64).reshape(B, 1).repeat(1, 4)  # stand-in "result ciphertexts"
    t0 = time.perf_counter()
    gathered = gather_results(local, total, dst=0) if world > 1 else local
    gather_ms = (time.perf_counter() - t0) * 1e3
    if rank == 0:
        assert gathered.shape[0] == total and bool((gathered[:, 0] == torch.arange(total)).all())
        print(json.dumps({"dry_run": True, "n_gpus": dist.get_world_size() if world > 1 else 1, "scaling": mode,
                          "total_batch": total, "batch_rank0": B, "gather_ms": gather_ms}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def run_rank(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.dry_run_cpu:
        return dry_run_cpu(args, rank, world)
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
    n_gpus = dist.get_world_size() if world > 1 else 1  # the ranks RCCL actually sees

    first, B, total, mode = my_share(args, rank, world)
    stream = torch.cuda.Stream(device=dev)  # inputs are generated on, and the C ABI launches into, this one stream
    primes = capi.create_primes(N, BITS)
    g = capi.Context(capi.CKKS, N, primes, device=local_rank)
    g.set_stream(stream.cuda_stream)
    g.keygen(0xABC00001)  # keys replicated on every device

    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)

    def rand_ct(chain, count):
        t = torch.empty((count, 2, L, N), dtype=torch.int64, device=dev)
        for j, q in enumerate(chain[:L]):
            t[:, :, j, :] = torch.randint(0, q, (count, 2, N), dtype=torch.int64, device=dev, generator=gen)
        return t

    with torch.cuda.stream(stream):
        a, b = rand_ct(primes, B), rand_ct(primes, B)  # uniformly random residues = what ciphertexts look like
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
    gather_ms = None
    if world > 1:
        tt = torch.tensor([elapsed, ev_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, ev_ms_max = float(tt[0]), float(tt[1])
        # the path's only exchange: gather the result ciphertexts of the last step on rank 0 (outside the timed region)
        from abc_amd.sharding import gather_results
        torch.cuda.synchronize()
        dist.barrier()
        tg = time.perf_counter()
        gathered = gather_results(out, total, dst=0)
        torch.cuda.synchronize()
        dist.barrier()
        gather_ms = (time.perf_counter() - tg) * 1e3
        if rank == 0:
            assert gathered.shape[0] == total
        del gathered

    if rank == 0:
        value = total * args.steps / elapsed
        launch_ms = ev_ms / args.steps  # one hot launch = one abc_hip_mul_relin over this rank's B pairs
        achieved = (ALGO_BYTES * B) / (launch_ms * 1e-3) / 1e9
        traffic, traffic_source = measured_traffic(B)
        line = {
            "metric": "homomorphic mul+relin/sec, CKKS N=2^14", "value": value, "unit": "mul+relin/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": mode, "vs_baseline": None,
            "dtype": "u64 (transforms: exact integer arithmetic in f64)", "data": "synthetic",
            "config": {"workload": "CKKS N=16384, 4 data limbs {50,40,40,40} + special 50-bit prime, ct x ct multiply + relinearize",
                       "total_batch": total, "batch_rank0": B,
                       "sharding": "independent ciphertext pairs per rank, result gather only"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         # SURVEY.md section 8d's second accounting: relin key counted as cache-resident across the batch (3 MiB / op)
                         "frac_key_resident": achieved / HBM_PEAK_GBS * ALGO_BYTES_KEY_RESIDENT / ALGO_BYTES,
                         "algorithmic_bytes_per_unit": ALGO_BYTES, "algorithmic_bytes_per_unit_key_resident": ALGO_BYTES_KEY_RESIDENT,
                         "kernel": "abc_hip_mul_relin (rank 0's launch) = k_split2_tensor_pass0_fp + k_split_special_fp (special prime) + "
                                   "k_split3_pass_fp + k_split4_main_fp (dominant), DESIGN.md section 4",
                         "algorithmic_bytes_per_launch": ALGO_BYTES * B, "launch_ms": launch_ms},
        }
        if gather_ms is not None:
            line["gather_ms"] = gather_ms
        # parity spot check of the measured path against the oracle: first / chunk-boundary / last pairs
        from oracle import oracle_py as om
        o = om.Oracle(om.CKKS, N, primes)
        o.keygen(0xABC00001)
        checked = sorted({i for i in (0, 255, 256, B - 1) if 0 <= i < B})
        for i in checked:
            ha, hb, ho = (t[i].cpu().numpy().view(np.uint64) for t in (a, b, out))
            if not np.array_equal(o.mul_relin(ha, hb), ho):
                raise SystemExit("bench: GPU result of pair %d differs from the oracle -- number is invalid" % i)
        line["checked_pairs_vs_oracle"] = checked
        if world == 1 and not args.no_60bit:
            # the headline chain keeps every key prime below 2^50 (fp64 butterflies); SEAL-typical 60-bit primes run the
            # integer kernels instead -- reported beside it so the precondition of `value` is visible
            del a, b, out
            p60 = capi.create_primes(N, BITS_60)
            g60 = capi.Context(capi.CKKS, N, p60, device=local_rank)
            g60.set_stream(stream.cuda_stream)
            g60.keygen(0xABC00001)
            B60 = min(B, 1024)
            with torch.cuda.stream(stream):
                a6, b6 = rand_ct(p60, B60), rand_ct(p60, B60)
                o6 = torch.empty_like(a6)
            p6 = [C.c_void_p(t.data_ptr()) for t in (a6, b6, o6)]
            for _ in range(3):
                g60.op("mul_relin", *p6, L, C.c_size_t(B60))
            torch.cuda.synchronize()
            t6 = time.perf_counter()
            for _ in range(10):
                g60.op("mul_relin", *p6, L, C.c_size_t(B60))
            torch.cuda.synchronize()
            v60 = 10 * B60 / (time.perf_counter() - t6)
            line["value_60bit_primes"] = v60
            line["value_60bit_primes_detail"] = {"value": v60, "unit": "mul+relin/s", "frac": v60 * ALGO_BYTES / (HBM_PEAK_GBS * 1e9),
                                                 "workload": "CKKS N=16384 {60,40,40,40 | 60}: integer kernels for the 60-bit primes, batch %d" % B60}
            o60 = om.Oracle(om.CKKS, N, p60)
            o60.keygen(0xABC00001)
            if not np.array_equal(o60.mul_relin(a6[B60 - 1].cpu().numpy().view(np.uint64), b6[B60 - 1].cpu().numpy().view(np.uint64)),
                                  o6[B60 - 1].cpu().numpy().view(np.uint64)):
                raise SystemExit("bench: 60-bit chain result differs from the oracle")
            g60.close()
        if world == 1 and not args.no_bfv:
            # the reference itself is BFV on BFVDefault(16384) with Batching(N, 20) (ref:src/runtime/SealCiphertextFactory.cpp:72-100):
            # the same operation on ITS scheme and ring, beside the CKKS headline (BEHZ multiply + relinearise, 8 data limbs)
            gb = capi.Context.bfv_default(16384, device=local_rank)
            gb.set_stream(stream.cuda_stream)
            gb.keygen(0xABC00001)
            Bb, Lb, nb = 256, gb.L, 16384
            with torch.cuda.stream(stream):
                def rand_bfv():
                    t = torch.empty((Bb, 2, Lb, nb), dtype=torch.int64, device=dev)
                    for j, q in enumerate(gb.primes[:Lb]):
                        t[:, :, j, :] = torch.randint(0, q, (Bb, 2, nb), dtype=torch.int64, device=dev)
                    return t
                ab, bb = rand_bfv(), rand_bfv()
                ob = torch.empty_like(ab)
            pb = [C.c_void_p(t.data_ptr()) for t in (ab, bb, ob)]
            for _ in range(3):
                gb.op("mul_relin", *pb, Lb, C.c_size_t(Bb))
            torch.cuda.synchronize()
            gb.timer_start()  # HIP events on the launch stream, as for the headline
            tb = time.perf_counter()
            for _ in range(10):
                gb.op("mul_relin", *pb, Lb, C.c_size_t(Bb))
            ev_b = gb.timer_stop()
            torch.cuda.synchronize()
            wall_b = time.perf_counter() - tb
            vb = 10 * Bb / wall_b
            algo_b = 8 * nb * (6 * Lb + 2 * Lb * (Lb + 1))  # SURVEY.md section 8d's formula at L = 8: 25.2 MB
            line["value_bfv_default_ring"] = {"value": vb, "unit": "mul+relin/s", "frac": vb * algo_b / (HBM_PEAK_GBS * 1e9),
                                              "algorithmic_bytes_per_unit": algo_b, "launch_ms": ev_b / 10, "steps": 10, "warmup": 3,
                                              "workload": "BFV BFVDefault(16384) (8 data limbs + special), t = Batching(16384, 20), batch 256"}
            ob_ = om.Oracle(om.BFV, nb, list(gb.primes), gb.t)
            ob_.keygen(0xABC00001)
            i = Bb - 1
            if not np.array_equal(ob_.mul_relin(ab[i].cpu().numpy().view(np.uint64), bb[i].cpu().numpy().view(np.uint64)),
                                  ob[i].cpu().numpy().view(np.uint64)):
                raise SystemExit("bench: BFV default-ring result differs from the oracle")
            del ab, bb, ob
            gb.close()
        if not args.no_cpu and world == 1:  # CPU leg on rank 0 at N=1 only
            line["cpu_baseline"] = cpu_baseline(primes, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        # nothing above this line has imported torch or touched HIP: the ranks are fresh processes
        sys.exit(launch_ranks(args))
    run_rank(args)


if __name__ == "__main__":
    main()
