#!/usr/bin/env python3
"""Interleaved A/B timing of the hot call under different ABC_HIP_* path switches, in ONE process on ONE device
(cdna_hip_programming.md rule 24): variants x rounds, median and min of the per-round rates.

  python tools/ab_hot.py --variants "default;ABC_HIP_NO_SPLIT4=1;ABC_HIP_LANES=1" --op mul_relin --batch 1024
"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N, L = 16384, 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="default;ABC_HIP_NO_SPLIT4=1")
    ap.add_argument("--op", default="mul_relin", choices=["mul_relin", "rotate"])
    ap.add_argument("--bits", default="50,40,40,40,50")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import torch
    from abc_amd import capi
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    bits = [int(x) for x in args.bits.split(",")]
    nl = len(bits) - 1
    primes = capi.create_primes(N, bits)
    g = capi.Context(capi.CKKS, N, primes)
    g.set_stream(stream.cuda_stream)
    g.keygen(1)
    B = args.batch
    with torch.cuda.stream(stream):
        def rand_ct():
            t = torch.empty((B, 2, nl, N), dtype=torch.int64, device=dev)
            for j, q in enumerate(primes[:nl]):
                t[:, :, j, :] = torch.randint(0, q, (B, 2, N), dtype=torch.int64, device=dev)
            return t
        a, b = rand_ct(), rand_ct()
        out = torch.empty_like(a)
    pa, pb, po = (C.c_void_p(t.data_ptr()) for t in (a, b, out))

    def step():
        if args.op == "mul_relin":
            g.op("mul_relin", pa, pb, po, nl, C.c_size_t(B))
        else:
            g.op("rotate", pa, po, nl, 1, C.c_size_t(B))

    variants = []
    for v in args.variants.split(";"):
        env = {}
        if v != "default":
            for kv in v.split(","):
                k, val = kv.split("=")
                env[k] = val
        variants.append((v, env))
    touched = sorted({k for _, e in variants for k in e})
    rates = {v: [] for v, _ in variants}
    ref = None
    for rnd in range(args.rounds + 1):  # round 0 = warm-up (workspace growth, first launches)
        for name, env in variants:
            for k in touched:
                os.environ.pop(k, None)
            os.environ.update(env)
            g.reload_env()
            step()
            torch.cuda.synchronize()
            if rnd == 0:  # all variants must agree bit for bit
                if ref is None:
                    ref = out.clone()
                elif not torch.equal(ref, out):
                    raise SystemExit("variant %s: result differs from the first variant" % name)
                continue
            g.timer_start()
            for _ in range(args.steps):
                step()
            ms = g.timer_stop()
            rates[name].append(B * args.steps / (ms * 1e-3))
    res = {name: {"median_per_s": statistics.median(r), "max_per_s": max(r), "us_per_op": 1e6 / statistics.median(r)}
           for name, r in rates.items()}
    print(json.dumps({"op": args.op, "bits": bits, "batch": B, "rounds": args.rounds, "steps": args.steps, "variants": res}, indent=1))


if __name__ == "__main__":
    main()
