"""Issue-rate probes of the integer-multiply and fp64 pipes (abc_hip_microbench)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abc_amd import capi

NAMES = {0: "shoup_lazy_modmul", 1: "barrett_modmul", 2: "fp64_modmul", 3: "mul_wide_64x64", 4: "fp64_fma"}


def main():
    n = 16384
    g = capi.Context(capi.CKKS, n, capi.create_primes(n, [50, 40, 40, 40, 50]))
    iters = 4096
    lanes = 256 * 8 * 256
    res = {}
    for which, name in NAMES.items():
        ms = g.microbench(which, iters)
        ops = lanes * 4.0 * iters
        res[name] = {"ms": ms, "Gop_per_s": ops / ms / 1e6}
        print("%-20s %8.3f ms  %10.1f Gop/s" % (name, ms, ops / ms / 1e6), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/microbench.json", "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
