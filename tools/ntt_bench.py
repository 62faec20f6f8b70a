"""Stand-alone NTT throughput probe: `limbs` independent 2^logn-point transforms per launch."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abc_amd import capi


def main():
    logn = int(sys.argv[1]) if len(sys.argv) > 1 else 14
    limbs = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    idx = int(sys.argv[4]) if len(sys.argv) > 4 else 0  # key-level prime index: 0/4 are 50-bit, 1..3 40-bit
    n = 1 << logn
    primes = capi.create_primes(n, [50, 40, 40, 40, 50])
    g = capi.Context(capi.CKKS, n, primes)
    rng = np.random.default_rng(0)
    x = rng.integers(0, primes[idx], size=(limbs, n), dtype=np.uint64)
    buf = g.upload(x)
    for inverse in (False, True):
        name = "ntt_inverse" if inverse else "ntt_forward"
        g.op(name, buf.ptr, 0, idx, C.c_size_t(limbs))
        g.sync()
        g.timer_start()
        for _ in range(reps):
            g.op(name, buf.ptr, 0, idx, C.c_size_t(limbs))
        ms = g.timer_stop() / reps
        per = ms * 1e3 / limbs * 256  # us per transform per CU
        bfly = limbs * (n // 2) * logn / (ms * 1e-3) / 1e9
        print("%s q%d 2^%d x %d: %.3f ms/launch, %.1f us per transform-CU, %.0f Gbutterfly/s, %.0f GB/s" %
              (name, idx, logn, limbs, ms, per, bfly, 2 * limbs * n * 8 / (ms * 1e-3) / 1e9), flush=True)


if __name__ == "__main__":
    main()
