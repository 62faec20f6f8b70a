import time, sys
sys.path.insert(0, '.')
from abc_amd import capi
for n in (4096, 16384, 32768):
    g = capi.Context.bfv_default(n)
    g.sync()
    t0 = time.perf_counter(); g.keygen(1); g.sync(); t1 = time.perf_counter()
    print("BFVDefault(%d): keygen (sk, pk, relin, %d Galois keys) %.1f ms" % (n, len(g.galois_elts()), (t1 - t0) * 1e3), flush=True)
    g.close()
