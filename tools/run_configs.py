"""BASELINE.json configs 2-5 end to end on the product path only (device keygen, encode, encrypt, evaluate, decrypt,
decode; verification against the cleartext computation in numpy).  Independent circuits are sharded over ranks:
`python tools/run_configs.py --gpus N --config 4` starts the N ranks itself (abc_amd/launcher.py: fresh processes before
anything touches the GPU, exactly as bench.py does; under torch.distributed.run it runs as the rank it is told), per-circuit
results are gathered on rank 0 over RCCL.  Every line carries the reference's four phase timers in its own order and unit --
t_keygen, t_input_encryption, t_computation, t_decryption in ms (ref:examples/main.cpp:41, the CSV its benchmark workflow
uploads).  These are parity / plumbing cases, not the bench line (bench.py is).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def shard_range(total, rank, world):  # abc_amd/sharding.py's partition (that module imports torch; the launcher must not)
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


class Phases:
    """the reference's four phase timers (ref:examples/main.cpp:41), wall clock in ms around a device synchronise"""
    NAMES = ("t_keygen", "t_input_encryption", "t_computation", "t_decryption")

    def __init__(self):
        self.ms = {k: 0.0 for k in self.NAMES}

    def run(self, name, g, fn):
        if g is not None:
            g.sync()
        t0 = time.perf_counter()
        r = fn()
        if g is not None:
            g.sync()
        self.ms[name] += (time.perf_counter() - t0) * 1e3
        return r


class Ckks:
    """Thin helper: batched CKKS ciphertext handles on one device."""

    def __init__(self, n, bits, device, seed=0xABC00001, ph=None):
        from abc_amd import capi
        t0 = time.perf_counter()
        self.n, self.primes = n, capi.create_primes(n, bits)
        self.g = capi.Context(capi.CKKS, n, self.primes, device=device)
        self.g.keygen(seed)
        self.g.sync()
        if ph is not None:
            ph.ms["t_keygen"] += (time.perf_counter() - t0) * 1e3
        self.L = self.g.L

    def encrypt(self, vectors, scale, seed):
        from abc_amd import ckks_encoder
        pl = np.stack([ckks_encoder.encode(v, scale, self.n, self.primes[: self.L]) for v in vectors])
        buf = self.g.upload(pl)
        self.g.op("ntt_limbs", buf.ptr, self.L, C.c_size_t(len(vectors)), 0)
        ct = self.g.alloc(len(vectors) * 2 * self.L * self.n * 8)
        self.g.op("encrypt", buf.ptr, C.c_uint64(seed), ct.ptr, C.c_size_t(len(vectors)))
        return ct

    def decrypt(self, ct, count, nl, scale):
        from abc_amd import ckks_encoder
        pl = self.g.alloc(count * nl * self.n * 8)
        self.g.op("decrypt", ct.ptr, 2, nl, pl.ptr, C.c_size_t(count))
        self.g.op("ntt_limbs", pl.ptr, nl, C.c_size_t(count), 1)
        res = self.g.download(pl, (count, nl, self.n))
        return [ckks_encoder.decode(r, scale, self.n, self.primes) for r in res]


GRAPH = {}  # config -> seconds of one recorded-circuit replay (abc_hip_graph_*: the whole circuit as ONE launch)


def timed_replay(g, circuit, cfg):
    """record the (already warmed-up) circuit once, replay it, time one replay"""
    try:
        g.graph_begin()
        circuit()
        gx = g.graph_end()
        g.graph_launch(gx); g.sync()
        t0 = time.perf_counter(); g.graph_launch(gx); g.sync()
        GRAPH[cfg] = time.perf_counter() - t0
        g.graph_destroy(gx)
    except Exception as e:  # the eager number stands; say why there is no replay figure
        GRAPH[cfg] = "not recorded: %s" % e


def config2(dev, rank, world, batch, ph):
    """BFV N=2^12, 2 limbs: ct x ct multiply + relinearize."""
    from abc_amd import capi
    n = 4096

    def setup():
        g_ = capi.Context.bfv_default(n, device=dev)
        g_.keygen(0xABC00001)
        return g_

    g = ph.run("t_keygen", None, setup)
    a0, a1 = shard_range(batch, rank, world)
    cnt = a1 - a0
    rng = np.random.default_rng(2)
    x = rng.integers(0, 1025, size=(batch, n))[a0:a1]  # the reference's value range, ref:test/end-to-end/BoxBlurTest.cpp:123
    y = rng.integers(0, 1025, size=(batch, n))[a0:a1]
    cb = C.c_size_t(cnt)

    def enc(v, seed):
        vb, pl, ct = g.upload(v.astype(np.int64)), g.alloc(cnt * n * 8), g.alloc(cnt * 2 * g.L * n * 8)
        g.op("batch_encode", vb.ptr, pl.ptr, cb)
        g.op("encrypt", pl.ptr, C.c_uint64(seed), ct.ptr, cb)
        return ct

    cx, cy = ph.run("t_input_encryption", g, lambda: (enc(x, 100 + a0), enc(y, 5000 + a0)))
    out = g.alloc(cnt * 2 * g.L * n * 8)
    g.op("mul_relin", cx.ptr, cy.ptr, out.ptr, g.L, cb); g.sync()  # warm-up: scratch arenas grow on first use
    t0 = time.perf_counter()
    ph.run("t_computation", g, lambda: g.op("mul_relin", cx.ptr, cy.ptr, out.ptr, g.L, cb))
    dt = time.perf_counter() - t0
    # the config as BASELINE.json words it -- ONE multiply + relinearise: median latency of the first pair alone
    one = C.c_size_t(1)
    lat = []
    for _ in range(21):
        t1 = time.perf_counter()
        g.op("mul_relin", cx.ptr, cy.ptr, out.ptr, g.L, one); g.sync()
        lat.append(time.perf_counter() - t1)
    config2.single_ms = sorted(lat)[len(lat) // 2] * 1e3
    g.op("mul_relin", cx.ptr, cy.ptr, out.ptr, g.L, cb); g.sync()

    def dec():
        pl, vals = g.alloc(cnt * n * 8), g.alloc(cnt * n * 8)
        g.op("decrypt", out.ptr, 2, g.L, pl.ptr, cb)
        g.op("batch_decode", pl.ptr, vals.ptr, cb)
        return g.download(vals, (cnt, n), np.int64)

    got = ph.run("t_decryption", g, dec)
    t = g.t
    want = (x * y) % t
    want = np.where(want > t // 2, want - t, want)
    return bool(np.array_equal(got, want)), cnt, dt, got[:, 0].astype(np.float64)


def config3(dev, rank, world, batch, ph):
    """CKKS N=2^14, 4 limbs: dot product (mul+relin, rescale, 13 x rotate+add)."""
    n, scale = 16384, 2.0 ** 40
    k = Ckks(n, [50, 40, 40, 40, 50], dev, ph=ph)
    a0, a1 = shard_range(batch, rank, world)
    cnt = a1 - a0
    rng = np.random.default_rng(3)
    xs, ys = rng.uniform(-1, 1, (batch, n // 2))[a0:a1], rng.uniform(-1, 1, (batch, n // 2))[a0:a1]
    g, cb = k.g, C.c_size_t(cnt)
    cx, cy = ph.run("t_input_encryption", g, lambda: (k.encrypt(xs, scale, 100 + a0), k.encrypt(ys, scale, 9000 + a0)))
    m, r, t = g.alloc(cnt * 2 * 4 * n * 8), g.alloc(cnt * 2 * 3 * n * 8), g.alloc(cnt * 2 * 3 * n * 8)

    def circuit():
        g.op("mul_relin", cx.ptr, cy.ptr, m.ptr, 4, cb)
        g.op("rescale", m.ptr, r.ptr, 2, 4, cb)
        step = n // 4
        while step >= 1:
            g.op("rotate", r.ptr, t.ptr, 3, step, cb)
            g.op("add", r.ptr, t.ptr, r.ptr, 2, 3, cb)
            step //= 2

    circuit(); g.sync()
    t0 = time.perf_counter(); ph.run("t_computation", g, circuit); dt = time.perf_counter() - t0
    if os.environ.get("ABC_CONFIGS_EAGER_AGAIN"):  # order check: the same eager pass once more BEFORE anything is recorded
        t1 = time.perf_counter(); circuit(); g.sync(); config3.eager_again = time.perf_counter() - t1
    timed_replay(g, circuit, 3)
    if os.environ.get("ABC_CONFIGS_EAGER_AGAIN"):  # ... and once AFTER the recording (allocator / plain caches as the recording left them)
        t1 = time.perf_counter(); circuit(); g.sync(); config3.eager_after_record = time.perf_counter() - t1
        t1 = time.perf_counter(); circuit(); g.sync(); config3.eager_after_record2 = time.perf_counter() - t1
    dec = ph.run("t_decryption", g, lambda: k.decrypt(r, cnt, 3, scale * scale / k.primes[3]))
    err = max(abs(d[0].real - float(np.dot(x, y))) for d, x, y in zip(dec, xs, ys))
    return bool(err < 1e-3), cnt, dt, np.array([d[0].real for d in dec])


def config4(dev, rank, world, batch, ph):
    """CKKS N=2^15: 8x8 box sum on 64x64 images (rotations 1,2,4,64,128,256 + adds)."""
    n, scale = 32768, 2.0 ** 30
    k = Ckks(n, [50, 40, 40, 50], dev, ph=ph)
    a0, a1 = shard_range(batch, rank, world)
    cnt = a1 - a0
    rng = np.random.default_rng(4)
    imgs = rng.integers(0, 1025, size=(batch, 64, 64)).astype(np.float64)[a0:a1]
    g, cb = k.g, C.c_size_t(cnt)
    ct = ph.run("t_input_encryption", g, lambda: k.encrypt(imgs.reshape(cnt, -1), scale, 100 + a0))
    acc, t = g.alloc(cnt * 2 * 3 * n * 8), g.alloc(cnt * 2 * 3 * n * 8)

    def circuit():
        g.op("memcpy_d2d", acc.ptr, ct.ptr, C.c_size_t(cnt * 2 * 3 * n * 8))
        for r in (1, 2, 4, 64, 128, 256):
            g.op("rotate", acc.ptr, t.ptr, 3, r, cb)
            g.op("add", acc.ptr, t.ptr, acc.ptr, 2, 3, cb)

    circuit(); g.sync()
    t0 = time.perf_counter(); ph.run("t_computation", g, circuit); dt = time.perf_counter() - t0
    timed_replay(g, circuit, 4)
    dec = ph.run("t_decryption", g, lambda: k.decrypt(acc, cnt, 3, scale))
    ok = True
    for d, im in zip(dec, imgs):
        ref = sum(np.roll(np.roll(im, -dx, axis=0), -dy, axis=1) for dx in range(8) for dy in range(8))
        ok = ok and np.abs(d.real[:4096].reshape(64, 64)[:56, :56] - ref[:56, :56]).max() < 1e-2
    return bool(ok), cnt, dt, np.array([d.real[0] for d in dec])


def config5(dev, rank, world, batch, ph, bits=None):
    """BFV N=2^16: depth-8 multiply chain."""
    from abc_amd import capi
    n = 65536
    # eight data limbs of ~55 bits + a special prime (explicit primes, no security table) -- integer kernels.
    # The same circuit on eight 49-bit data primes + a 50-bit special prime has enough budget for depth 8 at t = 20 bits (the
    # result is verified below) and keeps every prime below 2^50, so the exact-fp64 kernels apply: main() reports both.
    if bits is None:
        bits = int(os.environ.get("ABC_CONFIG5_BITS", "55"))
    t = capi.plain_modulus_batching(n, 20)

    def setup():
        primes = capi.create_primes(n, [bits] * 8 + [bits + 1])
        g_ = capi.Context(capi.BFV, n, primes, t, device=dev)
        g_.keygen(0xABC00001)
        return g_

    g = ph.run("t_keygen", None, setup)
    a0, a1 = shard_range(batch, rank, world)
    cnt = a1 - a0
    cb = C.c_size_t(cnt)
    rng = np.random.default_rng(5)
    vals = rng.integers(1, 8, size=(9, batch, n))[:, a0:a1]

    def enc_all():
        cts_ = []
        for j in range(9):
            vb, pl, ct = g.upload(vals[j].astype(np.int64)), g.alloc(cnt * n * 8), g.alloc(cnt * 2 * g.L * n * 8)
            g.op("batch_encode", vb.ptr, pl.ptr, cb)
            g.op("encrypt", pl.ptr, C.c_uint64(1000 * j + a0), ct.ptr, cb)
            cts_.append(ct)
        return cts_

    cts = ph.run("t_input_encryption", g, enc_all)
    acc = g.alloc(cnt * 2 * g.L * n * 8)

    def circuit():
        g.op("mul_relin", cts[0].ptr, cts[1].ptr, acc.ptr, g.L, cb)
        for j in range(2, 9):
            g.op("mul_relin", acc.ptr, cts[j].ptr, acc.ptr, g.L, cb)

    circuit(); g.sync()
    t0 = time.perf_counter(); ph.run("t_computation", g, circuit); dt = time.perf_counter() - t0

    def dec():
        pl, out = g.alloc(cnt * n * 8), g.alloc(cnt * n * 8)
        g.op("decrypt", acc.ptr, 2, g.L, pl.ptr, cb)
        g.op("batch_decode", pl.ptr, out.ptr, cb)
        return g.download(out, (cnt, n), np.int64) % t

    got = ph.run("t_decryption", g, dec)
    want = np.ones((cnt, n), dtype=np.int64)
    for j in range(9):
        want = (want * vals[j]) % t
    return bool(np.array_equal(got, want)), cnt, dt, got[:, 0].astype(np.float64)


CONFIGS = {2: (config2, 256), 3: (config3, 32), 4: (config4, 32), 5: (config5, 2)}


def dry_run_cpu(args, rank, world):
    """launch + sharding + result gather with gloo on the CPU: no device call, no numbers claimed (tests/test_bench_launcher.py)"""
    import torch
    import torch.distributed as dist
    from abc_amd.sharding import gather_results
    if world > 1:
        dist.init_process_group("gloo")
    for cfg in args.config:
        batch = args.batch or CONFIGS[cfg][1] * world
        a0, a1 = shard_range(batch, rank, world)
        local = torch.arange(a0, a1, dtype=torch.float64)  # stand-in per-circuit results
        got = gather_results(local, batch, dst=0) if world > 1 else local
        if rank == 0:
            assert got.shape[0] == batch and bool((got == torch.arange(batch, dtype=torch.float64)).all())
            print(json.dumps({"dry_run": True, "config": cfg, "n_gpus": dist.get_world_size() if world > 1 else 1,
                              "total_circuits": batch, "circuits_rank0": a1 - a0}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, nargs="*", default=[2, 3, 4, 5])
    ap.add_argument("--batch", type=int, default=0, help="independent circuits in total (default: per-config, per GPU)")
    ap.add_argument("--gpus", type=int, default=1, help="start this many ranks (one per GPU) from this script itself")
    ap.add_argument("--dry-run-cpu", action="store_true", help="rehearse launch + sharding + gather with gloo on the CPU")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        # nothing above this line has imported torch or touched HIP: the ranks are fresh processes (abc_amd/launcher.py)
        from abc_amd.launcher import launch_ranks
        sys.exit(launch_ranks(__file__, sys.argv[1:], args.gpus))
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_run_cpu:
        return dry_run_cpu(args, rank, world)
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(dev)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    for cfg in args.config:
        fn, default_batch = CONFIGS[cfg]
        batch = args.batch or default_batch * world
        ph = Phases()
        ok, cnt, dt, digest = fn(dev, rank, world, batch, ph)
        line = {"config": cfg, "rank": rank, "circuits": cnt, "verified": ok, "seconds": dt, "circuits_per_s": cnt / dt}
        line.update({k: round(v, 3) for k, v in ph.ms.items()})  # ms, the reference's CSV columns (ref:examples/main.cpp:41)
        if cfg in GRAPH:  # the same circuit recorded once and replayed as one graph launch (verified: the decrypted result is the replay's)
            line["recorded_replay_circuits_per_s"] = cnt / GRAPH[cfg] if isinstance(GRAPH[cfg], float) else GRAPH[cfg]
        if cfg == 2:
            line["single_mul_relin_ms"] = getattr(config2, "single_ms", None)
        if cfg == 3 and hasattr(config3, "eager_again"):
            line["eager_again_circuits_per_s"] = cnt / config3.eager_again
            line["eager_after_recording_circuits_per_s"] = cnt / config3.eager_after_record
            line["eager_after_recording_second_pass_circuits_per_s"] = cnt / config3.eager_after_record2
        if cfg == 5:
            line["chain"] = "8 x %s-bit + special" % os.environ.get("ABC_CONFIG5_BITS", "55")
            if "ABC_CONFIG5_BITS" not in os.environ:  # the parameter choice this backend recommends, beside the survey's
                ph49 = Phases()
                ok49, cnt49, dt49, _ = config5(dev, rank, world, batch, ph49, bits=49)
                line["fp64_chain_8x49bit"] = {"verified": ok49, "seconds": dt49, "circuits_per_s": cnt49 / dt49,
                                              **{k: round(v, 3) for k, v in ph49.ms.items()}}
                ok = ok and ok49
        if world > 1:
            import torch
            import torch.distributed as dist
            from abc_amd.sharding import gather_results
            tt = torch.tensor([float(ok), dt], dtype=torch.float64, device="cuda")
            mn = tt.clone(); dist.all_reduce(mn, op=dist.ReduceOp.MIN)
            mx = tt.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            # the path's only exchange: per-circuit results to rank 0 (RCCL point-to-point into one preallocated tensor)
            res = gather_results(torch.from_numpy(np.ascontiguousarray(digest)).to("cuda"), batch, dst=0)
            line.update({"all_verified": bool(mn[0] > 0.5), "total_circuits": batch, "total_circuits_per_s": batch / float(mx[1]),
                         "n_gpus": dist.get_world_size(), "gathered_results": int(res.shape[0]) if rank == 0 else None})
        if rank == 0:
            print(json.dumps(line), flush=True)
        if not ok:
            raise SystemExit("config %d: verification failed on rank %d" % (cfg, rank))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
