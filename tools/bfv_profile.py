import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from abc_amd import capi
n = int(sys.argv[1]); B = int(sys.argv[2])
g = capi.Context.bfv_default(n); g.keygen(1); L = g.L
rng = np.random.default_rng(0)
def rand_ct():
    x = np.stack([rng.integers(0, q, size=(B, 2, n), dtype=np.uint64) for q in g.primes[:L]], axis=2)
    return g.upload(x), x.nbytes
a, nb = rand_ct(); b, _ = rand_ct(); out = g.alloc(nb)
for _ in range(5):
    g.op("mul_relin", a.ptr, b.ptr, out.ptr, L, C.c_size_t(B))
g.sync()
