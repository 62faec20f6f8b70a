"""Per-operation throughput of the C ABI on one MI355X (batched, operands resident in HBM).
Not the bench line (bench.py is); numbers quoted in DESIGN.md / README.md come from here."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abc_amd import capi


def timeit(g, fn, reps=10):
    fn(); g.sync()
    g.timer_start()
    for _ in range(reps):
        fn()
    return g.timer_stop() / reps


def rand_ct(g, rng, batch, size, nl):
    x = np.stack([rng.integers(0, q, size=(batch, size, g.n), dtype=np.uint64) for q in g.primes[:nl]], axis=2)
    return g.upload(x), x.nbytes


def main():
    res = {}
    rng = np.random.default_rng(0)
    # ---- CKKS N=2^14, 4 limbs ----
    n, B = 16384, 512
    g = capi.Context(capi.CKKS, n, capi.create_primes(n, [50, 40, 40, 40, 50]))
    g.keygen(1)
    a, nb = rand_ct(g, rng, B, 2, 4)
    b, _ = rand_ct(g, rng, B, 2, 4)
    out = g.alloc(nb)
    cb = C.c_size_t(B)
    ops = {
        "ckks14_mul_relin": lambda: g.op("mul_relin", a.ptr, b.ptr, out.ptr, 4, cb),
        "ckks14_rotate_1": lambda: g.op("rotate", a.ptr, out.ptr, 4, 1, cb),
        "ckks14_rotate_3(NAF:2 key switches)": lambda: g.op("rotate", a.ptr, out.ptr, 4, 3, cb),
        "ckks14_add": lambda: g.op("add", a.ptr, b.ptr, out.ptr, 2, 4, cb),
        "ckks14_rescale": lambda: g.op("rescale", a.ptr, out.ptr, 2, 4, cb),
    }
    for k, fn in ops.items():
        ms = timeit(g, fn)
        res[k] = {"batch": B, "ms": ms, "ops_per_s": B / ms * 1e3}
        print("%-40s %8.3f ms / %d  -> %10.0f op/s" % (k, ms, B, B / ms * 1e3), flush=True)
    del a, b, out
    g.close()
    # ---- BFV N=2^12 (config 2) and BFVDefault(16384) ----
    for n, B in ((4096, 1024), (16384, 256)):
        g = capi.Context.bfv_default(n)
        g.keygen(1)
        L = g.L
        a, nb = rand_ct(g, rng, B, 2, L)
        b, _ = rand_ct(g, rng, B, 2, L)
        out = g.alloc(nb)
        cb = C.c_size_t(B)
        plain = g.upload(rng.integers(0, g.t, size=n, dtype=np.uint64))
        ops = {
            "bfv%d_mul_relin" % n: lambda: g.op("mul_relin", a.ptr, b.ptr, out.ptr, L, cb),
            "bfv%d_rotate_1" % n: lambda: g.op("rotate", a.ptr, out.ptr, L, 1, cb),
            "bfv%d_add" % n: lambda: g.op("add", a.ptr, b.ptr, out.ptr, 2, L, cb),
            "bfv%d_multiply_plain" % n: lambda: g.op("multiply_plain", a.ptr, plain.ptr, C.c_size_t(0), out.ptr, 2, L, cb),
        }
        for k, fn in ops.items():
            ms = timeit(g, fn)
            res[k] = {"batch": B, "ms": ms, "ops_per_s": B / ms * 1e3}
            print("%-40s %8.3f ms / %d  -> %10.0f op/s" % (k, ms, B, B / ms * 1e3), flush=True)
        # single-ciphertext latency (what the C++ plugin shim sees)
        one = C.c_size_t(1)
        ms = timeit(g, lambda: g.op("mul_relin", a.ptr, b.ptr, out.ptr, L, one), reps=20)
        res["bfv%d_mul_relin_latency_ms" % n] = ms
        print("%-40s %8.3f ms (batch 1)" % ("bfv%d_mul_relin latency" % n, ms), flush=True)
        # the same call replayed from a captured HIP graph (abc_hip_graph_*)
        g.op("mul_relin", a.ptr, b.ptr, out.ptr, L, one)  # scratch arenas sized before capture
        g.sync()
        g.graph_begin()
        g.op("mul_relin", a.ptr, b.ptr, out.ptr, L, one)
        ge = g.graph_end()
        ms = timeit(g, lambda: g.graph_launch(ge), reps=20)
        g.graph_destroy(ge)
        res["bfv%d_mul_relin_graph_latency_ms" % n] = ms
        print("%-40s %8.3f ms (batch 1, HIP graph replay)" % ("bfv%d_mul_relin latency" % n, ms), flush=True)
        del a, b, out
        g.close()
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/op_bench.json", "w"), indent=1)


if __name__ == "__main__":
    main()
