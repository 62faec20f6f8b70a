"""Self-consistency of the oracle's building blocks (so a golden-vector pass is not an accident)."""
import numpy as np
import pytest


def test_ntt_is_negacyclic_convolution(oracle_mod):
    n = 1024
    primes = oracle_mod.create_primes(n, [40, 41])
    o = oracle_mod.Oracle(oracle_mod.CKKS, n, primes)
    q = primes[0]
    rng = np.random.default_rng(0)
    a = rng.integers(0, q, n, dtype=np.uint64)
    b = rng.integers(0, q, n, dtype=np.uint64)
    fa, fb = o.ntt(0, a), o.ntt(0, b)
    prod = np.array([(int(x) * int(y)) % q for x, y in zip(fa, fb)], dtype=np.uint64)
    c = o.intt(0, prod)
    # schoolbook negacyclic product
    ref = [0] * n
    ai, bi = [int(v) for v in a], [int(v) for v in b]
    for i in range(n):
        if ai[i] == 0:
            continue
        for j in range(n):
            k = i + j
            v = ai[i] * bi[j]
            if k >= n:
                ref[k - n] -= v
            else:
                ref[k] += v
    assert [int(v) for v in c] == [v % q for v in ref]
    assert np.array_equal(o.intt(0, fa), a)


def test_ntt_root_is_minimal_primitive(oracle_mod):
    n = 4096
    o = oracle_mod.Oracle.bfv_default(n)
    for i, q in enumerate(o.primes):
        psi = int(oracle_mod.lib().orc_ctx_ntt_root(o.h, i))
        assert pow(psi, n, q) == q - 1
        # minimal among all primitive 2N-th roots (odd powers of psi)
        sq, cur, best = psi * psi % q, psi, psi
        for _ in range(n):
            best = min(best, cur)
            cur = cur * sq % q
        assert best == psi


def test_keyswitch_preserves_plaintext(oracle_mod):
    o = oracle_mod.Oracle.bfv_default(4096)
    o.keygen(5)
    vals = list(range(-20, 20))
    ct = o.encrypt(o.encode(oracle_mod.expand_vector(vals, o.n)), 9)
    sq = o.multiply(ct, ct)
    assert list(o.decode(o.decrypt(sq))[:40]) == [v * v for v in vals]          # size-3 decrypt
    assert list(o.decode(o.decrypt(o.relinearize(sq)))[:40]) == [v * v for v in vals]


def test_naf_rotation_equals_composed_rotations(oracle_mod):
    o = oracle_mod.Oracle.bfv_default(4096)
    o.keygen(6)
    vals = list(range(100))
    ct = o.encrypt(o.encode(oracle_mod.expand_vector(vals, o.n)), 10)
    # 7 = 8 - 1 in NAF: no direct key, so the result must match the slot semantics anyway
    d = o.decode(o.decrypt(o.rotate(ct, 7)))
    assert list(d[:93]) == vals[7:]
    assert o.elt_from_step(7) not in o.galois_elts()


def test_ckks_pipeline_accuracy(oracle_mod):
    n = 4096
    primes = oracle_mod.create_primes(n, [50, 40, 40, 50])
    o = oracle_mod.Oracle(oracle_mod.CKKS, n, primes)
    o.keygen(3)
    rng = np.random.default_rng(1)
    x, y = rng.uniform(-1, 1, n // 2), rng.uniform(-1, 1, n // 2)
    s = 2.0 ** 40
    cx, cy = o.encrypt(o.ckks_encode(x, s), 1), o.encrypt(o.ckks_encode(y, s), 2)
    m = o.rescale(o.mul_relin(cx, cy))
    got = o.ckks_decode(o.decrypt(m), s * s / primes[2]).real
    assert np.abs(got - x * y).max() < 1e-6   # CKKS is approximate: tolerance, not bit parity
    r = o.ckks_decode(o.decrypt(o.rotate(cx, 5)), s).real
    assert np.abs(r - np.roll(x, -5)).max() < 1e-5


def test_sampler_spec_is_deterministic(oracle_mod):
    a = oracle_mod.Oracle.bfv_default(4096)
    b = oracle_mod.Oracle.bfv_default(4096)
    a.keygen(11)
    b.keygen(11)
    assert np.array_equal(a.relin_key(), b.relin_key())
    b.keygen(12)
    assert not np.array_equal(a.relin_key(), b.relin_key())


def test_bfv_multiply_does_not_depend_on_the_auxiliary_base(oracle_mod):
    """BEHZ: every step of the multiply computes residues of one fixed integer, so the q-residues of the result must be the
    same whichever auxiliary primes carry the intermediate values.  SEAL draws m_sk and B from the 61-bit primes; the HIP
    backend draws them from the 50-bit ones where that puts every BEHZ transform on its exact-fp64 path.  Here: the oracle
    with ORC_BEHZ_AUX_BITS = 50 / 45 against the oracle with SEAL's base, on ordinary and on adversarial ciphertexts."""
    import os
    rng = np.random.default_rng(61)
    for n in (4096, 8192):
        primes, t = oracle_mod.default_bfv_primes(n), oracle_mod.plain_modulus_batching(n, 20)
        os.environ.pop("ORC_BEHZ_AUX_BITS", None)
        ref = oracle_mod.Oracle(oracle_mod.BFV, n, primes, t)
        ref.keygen(3)
        L = len(primes) - 1
        a = ref.encrypt(ref.encode(oracle_mod.expand_vector([3, 3, 1, 4, 5, 9], n)), 1)
        b = ref.encrypt(ref.encode(oracle_mod.expand_vector([0, 1, 2, 1, 10, 21], n)), 2)
        ex = np.stack([rng.integers(0, q, size=(2, n), dtype=np.uint64) for q in primes[:L]], axis=1)
        for j in range(L):
            ex[0, j, : n // 2] = primes[j] - 1
            ex[1, j, ::2] = 0
        want = [ref.multiply(a, b), ref.multiply(ex, ex), ref.multiply(a, ex)]
        for bits in (50, 45):
            os.environ["ORC_BEHZ_AUX_BITS"] = str(bits)
            try:
                alt = oracle_mod.Oracle(oracle_mod.BFV, n, primes, t)
            finally:
                os.environ.pop("ORC_BEHZ_AUX_BITS", None)
            got = [alt.multiply(a, b), alt.multiply(ex, ex), alt.multiply(a, ex)]
            for w, g_ in zip(want, got):
                assert np.array_equal(w, g_), (n, bits)


def test_timed_mul_relin_bfv_default_ring_runs_in_its_arena(oracle_mod):
    """orc_time_mul_relin (bench.py's cpu_baseline leg) bump-allocates every temporary of one multiply from an arena: BFV on
    BFVDefault(16384) needs ~300 limbs of it (the arena is sized from the scheme, and falls back to the heap when short)."""
    o = oracle_mod.Oracle.bfv_default(16384)
    o.keygen(5)
    a = o.encrypt(o.encode(oracle_mod.expand_vector([3, 3, 1, 4, 5, 9], o.n)), 1)
    b = o.encrypt(o.encode(oracle_mod.expand_vector([0, 1, 2, 1, 10, 21], o.n)), 2)
    assert o.time_mul_relin(a, b, 1) > 0.0
    assert list(o.decode(o.decrypt(o.mul_relin(a, b)))[:6]) == [0, 3, 2, 4, 50, 189]
