"""BFVDefault(n) mul+relin / rotate throughput against the batch size (tail effects of the per-chunk launches)."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abc_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
g = capi.Context.bfv_default(n); g.keygen(1); L = g.L
rng = np.random.default_rng(0)
for B in [int(x) for x in (sys.argv[2:] or ["32", "64", "128", "256", "512"])]:
    x = np.stack([rng.integers(0, q, size=(B, 2, n), dtype=np.uint64) for q in g.primes[:L]], axis=2)
    a, b, out = g.upload(x), g.upload(x[::-1].copy()), g.alloc(x.nbytes)
    cb = C.c_size_t(B)
    for name, fn in (("mul_relin", lambda: g.op("mul_relin", a.ptr, b.ptr, out.ptr, L, cb)),
                     ("rotate", lambda: g.op("rotate", a.ptr, out.ptr, L, 1, cb))):
        fn(); g.sync()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(5):
                fn()
            g.sync()
            best = min(best, (time.perf_counter() - t0) / 5)
        print("n=%d B=%4d %-10s %8.3f ms  %9.0f op/s" % (n, B, name, best * 1e3, B / best), flush=True)
    a.free(); b.free(); out.free()
