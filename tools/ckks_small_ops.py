import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from abc_amd import capi
n=16384
for bits in ([50,40,40,40,50],[60,40,40,40,60]):
    primes = capi.create_primes(n, bits); L=len(bits)-1
    g = capi.Context(capi.CKKS, n, primes); g.keygen(1)
    B=512
    rng=np.random.default_rng(0)
    x=np.stack([rng.integers(0,q,size=(B,2,n),dtype=np.uint64) for q in primes[:L]],axis=2)
    a=g.upload(x); out=g.alloc(x.nbytes); pl=g.upload(x[0,0].copy())
    cb=C.c_size_t(B)
    ops={"rescale": lambda: g.op("rescale", a.ptr, out.ptr, 2, L, cb),
         "add": lambda: g.op("add", a.ptr, a.ptr, out.ptr, 2, L, cb),
         "multiply_plain": lambda: g.op("multiply_plain", a.ptr, pl.ptr, C.c_size_t(0), out.ptr, 2, L, cb),
         "mod_switch": lambda: g.op("mod_switch", a.ptr, out.ptr, 2, L, cb)}
    for name,fn in ops.items():
        fn(); g.sync(); t0=time.perf_counter()
        for _ in range(5): fn()
        g.sync(); dt=(time.perf_counter()-t0)/5
        print(bits, name, "%.0f op/s"%(B/dt), flush=True)
