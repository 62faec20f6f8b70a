#!/usr/bin/env python3
"""Interleaved A/B timing of one C-ABI op at a chosen ring / chain / batch under ABC_HIP_* switches (one process, one device).
  python tools/ab_ops.py --logn 15 --bits 50,40,40,50 --op rotate --batch 256 --variants "default;ABC_HIP_NO_GSPLIT=1" """
import argparse, ctypes as C, json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--logn", type=int, default=15)
    ap.add_argument("--bits", default="50,40,40,50")
    ap.add_argument("--op", default="rotate", choices=["mul_relin", "rotate"])
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--variants", default="default")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--bfv-default", action="store_true", help="BFV on BFVDefault(2^logn) instead of a CKKS chain (--bits ignored)")
    ap.add_argument("--bfv", action="store_true", help="BFV on the --bits chain (t = Batching(n, 20))")
    a = ap.parse_args()
    import torch
    from abc_amd import capi
    n = 1 << a.logn
    bits = [int(x) for x in a.bits.split(",")]
    nl = len(bits) - 1
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    if a.bfv_default:
        g = capi.Context.bfv_default(n)
        primes, nl = list(g.primes), g.L
    elif a.bfv:
        primes = capi.create_primes(n, bits)
        g = capi.Context(capi.BFV, n, primes, capi.plain_modulus_batching(n, 20))
    else:
        primes = capi.create_primes(n, bits)
        g = capi.Context(capi.CKKS, n, primes)
    g.set_stream(stream.cuda_stream)
    g.keygen(1)
    B = a.batch
    with torch.cuda.stream(stream):
        def rand_ct():
            t = torch.empty((B, 2, nl, n), dtype=torch.int64, device=dev)
            for j, q in enumerate(primes[:nl]):
                t[:, :, j, :] = torch.randint(0, q, (B, 2, n), dtype=torch.int64, device=dev)
            return t
        x, y = rand_ct(), rand_ct()
        out = torch.empty_like(x)
    px, py, po = (C.c_void_p(t.data_ptr()) for t in (x, y, out))

    def step():
        if a.op == "mul_relin":
            g.op("mul_relin", px, py, po, nl, C.c_size_t(B))
        else:
            g.op("rotate", px, po, nl, 1, C.c_size_t(B))

    variants = []
    for v in a.variants.split(";"):
        env = {}
        if v != "default":
            for kv in v.split(","):
                k, val = kv.split("=")
                env[k] = val
        variants.append((v, env))
    touched = sorted({k for _, e in variants for k in e})
    rates = {v: [] for v, _ in variants}
    ref = None
    for rnd in range(a.rounds + 1):
        for name, env in variants:
            for k in touched:
                os.environ.pop(k, None)
            os.environ.update(env)
            g.reload_env()
            step()
            torch.cuda.synchronize()
            if rnd == 0:
                if ref is None:
                    ref = out.clone()
                elif not torch.equal(ref, out):
                    raise SystemExit("variant %s: result differs from the first variant" % name)
                continue
            g.timer_start()
            for _ in range(a.steps):
                step()
            ms = g.timer_stop()
            rates[name].append(B * a.steps / (ms * 1e-3))
    print(json.dumps({"op": a.op, "logn": a.logn, "bits": bits, "batch": B,
                      "variants": {k: {"median_per_s": statistics.median(r), "us_per_op": 1e6 / statistics.median(r)} for k, r in rates.items()}}, indent=1))


if __name__ == "__main__":
    main()
