#!/usr/bin/env python3
"""Randomised differential check of the C ABI against the CPU oracle (test infrastructure: lives under tests/ because it
uses oracle/; tests/test_gpu_fuzz.py runs a short fixed-seed campaign, the command below a longer one).  Each round draws a scheme, ring, modulus chain, level, batch size, dispatcher switches and an operation,
feeds both sides the same random residues (uniform and end-of-range values -- parity is defined on residues, the inputs need
not be valid encryptions) and compares every output word.  The fixed-parameter tests cover the paths the dispatcher is known
to take; this walks the combinations nobody wrote down (levels 1..L at every ring, odd batches across chunk boundaries,
mixed prime sizes, rotations by arbitrary steps).

  python tests/fuzz_parity.py --seconds 60 --seed 1        (needs a GPU; exits 1 and prints the case on the first mismatch)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

SWITCH_SETS = [
    {}, {}, {}, {},  # the default path gets most of the draws
    {"ABC_HIP_CHUNK": "3", "ABC_HIP_LANES": "2"},
    {"ABC_HIP_CHUNK": "2", "ABC_HIP_LANES": "1"},
    {"ABC_HIP_NO_FP64": "1"},
    {"ABC_HIP_NO_MIXED": "1"},
    {"ABC_HIP_NO_ISPLIT": "1"},
    {"ABC_HIP_NO_SPLIT": "1"},
    {"ABC_HIP_NO_SPLIT4": "1"},
    {"ABC_HIP_NO_PACK": "1"},
    {"ABC_HIP_NO_KEY_TWIN": "1"},
    {"ABC_HIP_NO_BMUL": "1"},
    {"ABC_HIP_NO_SPECIAL8X2": "1"},
    {"ABC_HIP_NO_FINISH_LDS": "1"},
    {"ABC_HIP_NO_IKS": "1"},
    {"ABC_HIP_NO_GALOIS_FUSION": "1"},
    {"ABC_HIP_NO_BMUL_R6": "1"},
    {"ABC_HIP_NO_LEAN_FRONT": "1"},
    {"ABC_HIP_NO_FUSED": "1"},
    {"ABC_HIP_NO_BSPLIT": "1"},
    {"ABC_HIP_NO_GSPLIT": "1"},
    {"ABC_HIP_NO_TENSOR_INTT": "1"},
    {"ABC_HIP_BEHZ_SEAL_BASE": "1"},
    {"ABC_HIP_BFV_SCRATCH_MB": "64"},
    {"ABC_HIP_FEW_LIMBS": "0"},
]
ALL_SWITCHES = sorted({k for s in SWITCH_SETS for k in s})


def random_ct(rng, primes, nl, n, batch, size=2):
    ct = np.empty((batch, size, nl, n), dtype=np.uint64)
    for j in range(nl):
        q = int(primes[j])
        ct[:, :, j, :] = rng.integers(0, q, size=(batch, size, n), dtype=np.uint64)
        pool = np.array([0, 1, (q - 1) // 2, (q + 1) // 2, q - 2, q - 1], dtype=np.uint64)
        edge = rng.integers(0, 32, size=(batch, size, n))
        ct[:, :, j, :] = np.where(edge < 6, pool[np.minimum(edge, 5)], ct[:, :, j, :])
    if rng.integers(0, 4) == 0:  # a long run of q - 1: where lazy bounds break first
        ct[0, 0, :, : n // 4] = np.array([int(p) - 1 for p in primes[:nl]], dtype=np.uint64)[:, None]
    return ct


def draw_case(rng, max_logn):
    scheme = "bfv" if rng.integers(0, 2) else "ckks"
    logn = int(rng.choice([12, 12, 13, 13, 14, 14, 14, 15][: 8 if max_logn >= 15 else 7]))
    logn = min(logn, max_logn)
    n = 1 << logn
    case = {"scheme": scheme, "logn": logn, "switches": dict(SWITCH_SETS[int(rng.integers(0, len(SWITCH_SETS)))])}
    if scheme == "bfv":
        if rng.integers(0, 3) and logn <= 14:
            case["bits"] = None  # BFVDefault(n)
        else:
            L = int(rng.integers(1, 5 if logn < 15 else 4))
            b = int(rng.choice([36, 40, 45, 49, 50, 52, 55, 58]))
            case["bits"] = [b] * L + [min(b + 1, 60)]
    else:
        L = int(rng.integers(1, 8)) if rng.integers(0, 3) == 0 else int(rng.integers(1, 5))  # up to seven data limbs
        style = int(rng.integers(0, 4))
        if style == 0:
            bits = [50] + [40] * (L - 1) + [50]
        elif style == 1:
            bits = [60] + [40] * (L - 1) + [60]
        elif style == 2:
            bits = [int(rng.choice([45, 49, 50])) for _ in range(L + 1)]
        else:
            bits = [int(rng.choice([40, 50, 51, 55, 57, 60])) for _ in range(L + 1)]
        case["bits"] = bits
    case["batch"] = int(rng.choice([1, 1, 2, 3, 5, 7, 9]))
    ops = ["mul_relin", "rotate", "add", "sub", "negate", "multiply", "relinearize"]
    if scheme == "bfv":
        ops += ["multiply_plain", "add_plain", "sub_plain"]
    else:
        ops += ["rescale", "mod_switch"]
    case["op"] = str(rng.choice(ops))
    case["step"] = int(rng.integers(-(n // 2) + 1, n // 2)) if rng.integers(0, 2) else int(rng.choice([1, -1, 2, 3, 7, 64, -24, n // 4]))
    if case["step"] == 0:
        case["step"] = 1
    case["alias"] = int(rng.choice([0, 0, 1, 2]))
    case["level_draw"] = float(rng.random())
    case["data_seed"] = int(rng.integers(0, 2 ** 31))
    return case


def run_case(case, om, capi, contexts):
    for k in ALL_SWITCHES:
        os.environ.pop(k, None)
    os.environ.update(case["switches"])
    n = 1 << case["logn"]
    bfv = case["scheme"] == "bfv"
    key = (case["scheme"], case["logn"], tuple(case["bits"]) if case["bits"] else None, tuple(sorted(case["switches"].items())))
    if key not in contexts:
        if len(contexts) >= 6:  # keep device and host memory bounded
            for _, (o_, g_) in list(contexts.items())[:3]:
                g_.close()
            for k in list(contexts)[:3]:
                del contexts[k]
        if bfv:
            primes = om.default_bfv_primes(n) if case["bits"] is None else om.create_primes(n, case["bits"])
            t = om.plain_modulus_batching(n, 20)
            o = om.Oracle(om.BFV, n, primes, t)
            g = capi.Context(capi.BFV, n, primes, t)
        else:
            primes = om.create_primes(n, case["bits"])
            o = om.Oracle(om.CKKS, n, primes)
            g = capi.Context(capi.CKKS, n, primes)
        o.keygen(0xABC00001)
        g.keygen(0xABC00001)  # shared sampling spec: identical keys on both sides
        contexts[key] = (o, g)
    o, g = contexts[key]
    g.reload_env()
    L = len(o.primes) - 1
    nl = L if bfv else 1 + int(case["level_draw"] * L)
    nl = min(max(nl, 1), L)
    rng = np.random.default_rng(case["data_seed"])
    B = case["batch"]
    op = case["op"]
    primes = [int(p) for p in o.primes]
    if op == "relinearize":
        x = random_ct(rng, primes, nl, n, B, size=3)
        got, want = g.relinearize(x), np.stack([o.relinearize(r) for r in x])
    elif op in ("mul_relin", "multiply", "add", "sub"):
        x, y = random_ct(rng, primes, nl, n, B), random_ct(rng, primes, nl, n, B)
        alias = case.get("alias", 0) if op != "multiply" else 0
        if alias:  # the C ABI allows the result to overwrite an operand: 1 = first, 2 = second
            import ctypes as C
            bx, by = g.upload(x), g.upload(y)
            dst = bx if alias == 1 else by
            if op == "mul_relin":
                g.op(op, bx.ptr, by.ptr, dst.ptr, nl, C.c_size_t(B))
            else:
                g.op(op, bx.ptr, by.ptr, dst.ptr, 2, nl, C.c_size_t(B))
            got = g.download(dst, x.shape)
            bx.free(); by.free()
        else:
            got = getattr(g, op)(x, y)
        want = np.stack([getattr(o, op)(a, b) for a, b in zip(x, y)])
    elif op == "negate":
        x = random_ct(rng, primes, nl, n, B)
        got, want = g.negate(x), np.stack([o.negate(r) for r in x])
    elif op == "rotate":
        x = random_ct(rng, primes, nl, n, B)
        step = case["step"]
        if not bfv:
            step = step % (n // 2) or 1
        got, want = g.rotate(x, step), np.stack([o.rotate(r, step) for r in x])
    elif op in ("multiply_plain", "add_plain", "sub_plain"):
        x = random_ct(rng, primes, nl, n, B)
        plain = rng.integers(0, o.t, size=n, dtype=np.uint64)
        got = getattr(g, op)(x, plain)
        want = np.stack([getattr(o, op)(r, plain) for r in x])
    elif op in ("rescale", "mod_switch"):
        if nl < 2:
            return "skip"
        x = random_ct(rng, primes, nl, n, B)
        got, want = getattr(g, op)(x), np.stack([getattr(o, op)(r) for r in x])
    else:
        raise ValueError(op)
    if got.shape != want.shape or not np.array_equal(got, want):
        bad = np.argwhere(got != want) if got.shape == want.shape else []
        return "MISMATCH nl=%d: %d words differ%s" % (nl, len(bad), (", first at %s" % (tuple(int(v) for v in bad[0]),)) if len(bad) else "")
    return "ok"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-logn", type=int, default=14, help="15 adds N = 2^15 (the oracle then needs seconds per draw)")
    ap.add_argument("--max-cases", type=int, default=0)
    ap.add_argument("--case", default="", help="replay one case: the JSON object a failed campaign printed under \"case\"")
    ap.add_argument("--also", default="", help="with --case: extra switches for the replay, NAME=1,NAME=1")
    a = ap.parse_args()
    import oracle_py as om
    from abc_amd import capi
    if a.case:
        case = json.loads(a.case)
        for kv in filter(None, a.also.split(",")):
            case["switches"][kv.split("=")[0]] = kv.split("=")[1]
        print(json.dumps({"result": run_case(case, om, capi, {}), "case": case}), flush=True)
        return
    rng = np.random.default_rng(a.seed)
    contexts, counts = {}, {}
    t0 = last_note = time.time()
    done = 0
    while time.time() - t0 < a.seconds and (not a.max_cases or done < a.max_cases):
        case = draw_case(rng, a.max_logn)
        res = run_case(case, om, capi, contexts)
        done += 1
        tag = "%s/%s" % (case["scheme"], case["op"])
        counts[tag] = counts.get(tag, 0) + (res == "ok")
        if time.time() - last_note > 60:
            last_note = time.time()
            print("... %d cases, %.0f s" % (done, time.time() - t0), flush=True)
        if res.startswith("MISMATCH"):
            print(json.dumps({"result": res, "case": case}), flush=True)
            sys.exit(1)
    print(json.dumps({"cases": done, "seconds": round(time.time() - t0, 1), "passed_by_kind": counts}), flush=True)


if __name__ == "__main__":
    main()
