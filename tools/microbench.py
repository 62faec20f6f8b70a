"""Issue-rate probes of the integer-multiply and fp64 pipes (abc_hip_microbench).
Needs a library with the probe kernels: `python -m abc_amd.build --microbench` (the default build leaves them out)."""
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
    # raw instruction issue rates: 1024 workgroups x 4 waves, 8 instructions per iteration
    INSTR = {100: "v_mad_u64_u32", 101: "v_mul_lo_u32", 102: "v_mul_hi_u32", 103: "v_mul_u32_u24", 104: "v_add_u32",
             105: "v_lshl_add_u64", 106: "v_add_co_u32", 107: "v_cndmask_b32", 108: "v_mad_u32_u24", 109: "v_mul_hi_u32_u24",
             110: "v_fma_f64", 111: "v_mul_f64", 112: "v_add_f64", 113: "v_rndne_f64", 114: "v_cvt_f64_u32", 115: "v_and_b32",
             116: "v_pk_add_f32"}
    it = 8192
    for which, name in INSTR.items():
        ms = g.microbench(which, it)
        # wave-instructions per SIMD: 4 waves x it x 8; cycles at 2.4 GHz
        cyc = ms * 1e-3 * 2.4e9 / (4 * it * 8)
        res[name] = {"ms": ms, "cycles_per_wave_instr_at_2.4GHz": cyc}
        print("%-20s %8.3f ms  %6.2f cycles per wave-instruction (if 2.4 GHz)" % (name, ms, cyc), flush=True)
    # unguarded butterfly: compiler output vs a hand-scheduled 16-instruction sequence (4 butterflies per iteration)
    it = 4096
    for which, name in ((200, "butterfly_cpp"), (201, "butterfly_asm"), (202, "butterfly_fp64")):
        ms = g.microbench(which, it)
        cyc = ms * 1e-3 * 2.4e9 / (4 * it * 4)  # wave-butterflies per SIMD: 4 waves x it x 4
        res[name] = {"ms": ms, "cycles_per_wave_butterfly_at_2.4GHz": cyc, "Gbutterfly_per_s": 1024 * 256 * 4.0 * it / ms / 1e6}
        print("%-20s %8.3f ms  %6.1f cycles per wave-butterfly (if 2.4 GHz)  %8.0f Gbfly/s" % (name, ms, cyc, res[name]["Gbutterfly_per_s"]), flush=True)
    # fp64 forward transform, 4096 per launch, with the HBM load / store phases cut out
    for which, name in ((300, "ntt_fp_full"), (301, "ntt_fp_no_load"), (302, "ntt_fp_no_store"), (303, "ntt_fp_no_load_no_store"),
                        (304, "ntt_fp_persistent"), (305, "ntt_fp_persistent_prefetch"),
                        (306, "ntt_fp_split_adjacent"), (307, "ntt_fp_split_same_xcd"),
                        (308, "copy_transform_pattern"), (309, "copy_streaming"),
                        (310, "ntt_fp_stagger_6us"), (311, "ntt_fp_stagger_12us"), (312, "ntt_fp_stagger_17us"),
                        (313, "ntt4096x4_no_stagger"), (314, "ntt4096x4_stagger_2us"), (315, "ntt4096x4_stagger_4us"),
                        (316, "ntt4096x4_stagger_9us")):
        ms = g.microbench(which, 4096)
        res[name] = {"ms": ms, "us_per_transform_per_cu": ms * 1e3 / 4096 * 256}
        print("%-26s %8.3f ms  %6.2f us per transform per CU  (%.0f GB/s if 256 KiB per transform)" %
              (name, ms, ms * 1e3 / 4096 * 256, 4096 * 262144 / ms / 1e6), flush=True)
    # can arithmetic and HBM streaming overlap at chip level?  fp64 FMA kernel and 1 GiB-per-pass copy, alone / together
    for which, name in ((320, "fma_only"), (321, "copy_only"), (322, "fma_and_copy_concurrent")):
        ms = g.microbench(which, 8)
        res[name] = {"ms": ms}
        print("%-26s %8.3f ms" % (name, ms), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/microbench.json", "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
