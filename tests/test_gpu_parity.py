"""GPU parity: every HIP evaluator op, called through the C ABI (include/abc_hip.h), must return
residues bit-identical to the CPU oracle on the same inputs and keys.

Mirrors test/runtime/SealCiphertextFactoryTest.cpp of the reference (same operand vectors
{3,3,1,4,5,9} o {0,1,2,1,10,21}, N = 4096, :14,:19) and extends it to residue level.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

D1 = [3, 3, 1, 4, 5, 9]
D2 = [0, 1, 2, 1, 10, 21]
ROT = [123456, 3, 1, 4, 5, 9, 5, 2, 1, 5]


def _report(name, got, want):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, "%s: shape %s vs %s" % (name, got.shape, want.shape)
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)
        msg = "%s: %d/%d words differ; first at %s got %d want %d" % (
            name, len(bad), got.size, tuple(bad[0]), got[tuple(bad[0])], want[tuple(bad[0])])
        raise AssertionError(msg)


@pytest.fixture(scope="module")
def bfv(oracle_mod, capi):
    """config 2: BFV N=2^12, BFVDefault(4096) = 2 data limbs + special, t = 1032193."""
    o = oracle_mod.Oracle.bfv_default(4096)
    o.keygen(0xABC00001)
    g = capi.Context.bfv_default(4096)
    assert g.primes == o.primes and g.t == o.t
    g.load_keys(sk=o.secret_key(), pk=o.public_key(), relin=o.relin_key(),
                galois={e: o.galois_key(e) for e in o.galois_elts()})
    return o, g


def _enc(o, om, vals, seed):
    return o.encrypt(o.encode(om.expand_vector(vals, o.n)), seed)


def test_ntt_roundtrip_and_parity(bfv):
    o, g = bfv
    rng = np.random.default_rng(1)
    for idx, q in enumerate(o.primes):
        x = rng.integers(0, q, size=(3, o.n), dtype=np.uint64)
        f = g.ntt(x, 0, idx)
        want = np.stack([o.ntt(idx, r) for r in x])
        _report("ntt fwd prime %d" % idx, f, want)
        b = g.ntt(f, 0, idx, inverse=True)
        _report("ntt inv prime %d" % idx, b, x)


def test_add_sub_negate(bfv, oracle_mod):
    o, g = bfv
    a, b = _enc(o, oracle_mod, D1, 1), _enc(o, oracle_mod, D2, 2)
    _report("add", g.add(a, b), o.add(a, b))
    _report("sub", g.sub(a, b), o.sub(a, b))
    _report("negate", g.negate(a), o.negate(a))


def test_bfv_multiply_and_relin(bfv, oracle_mod):
    o, g = bfv
    a, b = _enc(o, oracle_mod, D1, 1), _enc(o, oracle_mod, D2, 2)
    m3 = o.multiply(a, b)
    _report("bfv multiply (BEHZ)", g.multiply(a, b), m3)
    _report("relinearize", g.relinearize(m3), o.relinearize(m3))
    r = g.mul_relin(a, b)
    _report("mul_relin", r, o.mul_relin(a, b))
    assert list(o.decode(o.decrypt(r))[:6]) == [0, 3, 2, 4, 50, 189]  # SealCiphertextFactoryTest.cpp:178-192


def test_rotate(bfv, oracle_mod):
    o, g = bfv
    ct = _enc(o, oracle_mod, ROT, 3)
    for steps in (4, -24, 6, 1, -1, 1024, 7):
        _report("rotate %d" % steps, g.rotate(ct, steps), o.rotate(ct, steps))


def test_plain_ops(bfv, oracle_mod):
    o, g = bfv
    a = _enc(o, oracle_mod, D1, 1)
    pl = o.encode(oracle_mod.expand_vector(D2, o.n))
    _report("multiply_plain", g.multiply_plain(a, pl), o.multiply_plain(a, pl))
    _report("add_plain", g.add_plain(a, pl), o.add_plain(a, pl))
    _report("sub_plain", g.sub_plain(a, pl), o.sub_plain(a, pl))


def test_encode_decode_encrypt_decrypt(bfv, oracle_mod):
    o, g = bfv
    vals = np.array(oracle_mod.expand_vector(D1 + [-7, 500000, -500000], o.n), dtype=np.int64)
    pl = o.encode(vals)
    _report("batch_encode", g.batch_encode(vals), pl)
    _report("batch_decode", g.batch_decode(pl), o.decode(pl))
    ct = o.encrypt(pl, 77)
    _report("encrypt (same sampling spec)", g.encrypt(pl, 77), ct)
    _report("decrypt", g.decrypt(ct), o.decrypt(ct))
    m3 = o.multiply(ct, ct)
    _report("decrypt size 3", g.decrypt(m3), o.decrypt(m3))


def test_batched_encryption_draws_on_several_host_threads(bfv, oracle_mod):
    """a batch of ciphertexts is sampled by up to 16 host threads (abc_keys.hip, encrypt_with): ciphertext i of a seeded call is
    still the oracle's encryption under seed + i, whatever the thread count; the OS-keyed path decrypts and never repeats itself"""
    o, g = bfv
    pls = np.stack([o.encode(np.array(oracle_mod.expand_vector([k + 1, -k, 3 * k], o.n), dtype=np.int64)) for k in range(21)])
    got = g.encrypt(pls, 5000)
    for k in (0, 1, 7, 13, 20):
        _report("seeded batch encrypt [%d]" % k, got[k], o.encrypt(pls[k], 5000 + k))
    sec = g.encrypt(pls, None)  # abc_hip_encrypt_secure: one ChaCha20 instance per worker, keyed by getrandom(2)
    assert sec.shape == got.shape
    dec = g.decrypt(sec)
    for k in range(21):
        assert np.array_equal(dec[k], pls[k]), "secure batch encrypt: ciphertext %d does not decrypt to its plaintext" % k
    flat = sec.reshape(21, -1)
    assert len({flat[k, :64].tobytes() for k in range(21)}) == 21  # independent randomness per ciphertext
    assert not np.array_equal(sec, g.encrypt(pls, None))


def test_keygen_matches_oracle(oracle_mod, capi):
    o = oracle_mod.Oracle.bfv_default(4096)
    o.keygen(424242)
    g = capi.Context.bfv_default(4096)
    g.keygen(424242)
    _report("sk", g.get_key("sk"), o.secret_key())
    _report("pk", g.get_key("pk"), o.public_key())
    _report("relin", g.get_key("relin"), o.relin_key())
    assert g.galois_elts() == o.galois_elts()
    for e in o.galois_elts()[:4]:
        _report("galois %d" % e, g.get_key("galois", e), o.galois_key(e))


def test_batched_ops_match_single(bfv, oracle_mod):
    o, g = bfv
    rng = np.random.default_rng(5)
    cts_a = np.stack([_enc(o, oracle_mod, list(rng.integers(0, 1024, 10)), 100 + i) for i in range(5)])
    cts_b = np.stack([_enc(o, oracle_mod, list(rng.integers(0, 1024, 10)), 200 + i) for i in range(5)])
    want = np.stack([o.mul_relin(a, b) for a, b in zip(cts_a, cts_b)])
    _report("batched mul_relin", g.mul_relin(cts_a, cts_b), want)
    want = np.stack([o.rotate(a, 3) for a in cts_a])
    _report("batched rotate", g.rotate(cts_a, 3), want)


# ---------------- CKKS (config 3 shape) ----------------
@pytest.fixture(scope="module")
def ckks(oracle_mod, capi):
    n = 16384
    primes = oracle_mod.create_primes(n, [50, 40, 40, 40, 50])
    assert capi.create_primes(n, [50, 40, 40, 40, 50]) == primes
    o = oracle_mod.Oracle(oracle_mod.CKKS, n, primes)
    o.keygen(0xABC00001)
    g = capi.Context(capi.CKKS, n, primes)
    g.load_keys(sk=o.secret_key(), pk=o.public_key(), relin=o.relin_key(),
                galois={e: o.galois_key(e) for e in o.galois_elts()})
    rng = np.random.default_rng(7)
    x, y = rng.uniform(-1, 1, n // 2), rng.uniform(-1, 1, n // 2)
    scale = 2.0 ** 40
    cx, cy = o.encrypt(o.ckks_encode(x, scale), 11), o.encrypt(o.ckks_encode(y, scale), 12)
    return o, g, x, y, cx, cy, scale


def test_ckks_ntt(ckks):
    o, g = ckks[0], ckks[1]
    rng = np.random.default_rng(2)
    for idx, q in enumerate(o.primes):
        x = rng.integers(0, q, size=(2, o.n), dtype=np.uint64)
        f = g.ntt(x, 0, idx)
        _report("ntt16384 fwd %d" % idx, f, np.stack([o.ntt(idx, r) for r in x]))
        _report("ntt16384 inv %d" % idx, g.ntt(f, 0, idx, inverse=True), x)


def test_ckks_mul_relin_rescale(ckks):
    o, g, x, y, cx, cy, scale = ckks
    m3 = o.multiply(cx, cy)
    _report("ckks multiply", g.multiply(cx, cy), m3)
    _report("ckks relinearize", g.relinearize(m3), o.relinearize(m3))
    mr = o.mul_relin(cx, cy)
    _report("ckks mul_relin", g.mul_relin(cx, cy), mr)
    rs = o.rescale(mr)
    _report("ckks rescale", g.rescale(mr), rs)
    _report("ckks mod_switch", g.mod_switch(mr), o.mod_switch(mr))
    # lower level key switch
    r2 = o.mul_relin(rs, rs)
    _report("ckks mul_relin level 3", g.mul_relin(rs, rs), r2)
    dec = o.ckks_decode(o.decrypt(g.rescale(g.mul_relin(cx, cy))), scale * scale / o.primes[3])
    assert np.abs(dec.real - x * y).max() < 1e-5  # CKKS tolerance (approximate scheme)


def test_ckks_rotate_and_plain(ckks):
    o, g, x, y, cx, cy, scale = ckks
    for steps in (1, -3, 5, 4096):
        _report("ckks rotate %d" % steps, g.rotate(cx, steps), o.rotate(cx, steps))
    pl = o.ckks_encode(y, scale)
    _report("ckks multiply_plain", g.multiply_plain(cx, pl), o.multiply_plain(cx, pl))
    _report("ckks add_plain", g.add_plain(cx, pl), o.add_plain(cx, pl))
    _report("ckks encrypt", g.encrypt(pl, 99), o.encrypt(pl, 99))
    _report("ckks decrypt", g.decrypt(cx), o.decrypt(cx))


# ---------------- large rings: the strided + block NTT path ----------------
@pytest.mark.parametrize("logn", [15, 16])
def test_big_ring_ntt(oracle_mod, capi, logn):
    n = 1 << logn
    primes = oracle_mod.create_primes(n, [55, 55, 56])
    o = oracle_mod.Oracle(oracle_mod.CKKS, n, primes)
    g = capi.Context(capi.CKKS, n, primes)
    rng = np.random.default_rng(3)
    for idx, q in enumerate(primes):
        x = rng.integers(0, q, size=(2, n), dtype=np.uint64)
        f = g.ntt(x, 0, idx)
        _report("ntt 2^%d fwd %d" % (logn, idx), f, np.stack([o.ntt(idx, r) for r in x]))
        _report("ntt 2^%d inv %d" % (logn, idx), g.ntt(f, 0, idx, inverse=True), x)


@pytest.mark.parametrize("logn", [10, 11, 13])
def test_other_ring_sizes_ntt(oracle_mod, capi, logn):
    n = 1 << logn
    primes = oracle_mod.create_primes(n, [40, 41])
    o = oracle_mod.Oracle(oracle_mod.CKKS, n, primes)
    g = capi.Context(capi.CKKS, n, primes)
    rng = np.random.default_rng(4)
    x = rng.integers(0, primes[0], size=(3, n), dtype=np.uint64)
    f = g.ntt(x, 0, 0)
    _report("ntt 2^%d fwd" % logn, f, np.stack([o.ntt(0, r) for r in x]))
    _report("ntt 2^%d inv" % logn, g.ntt(f, 0, 0, inverse=True), x)


# ---------------- both kernel paths: LDS-resident (default, N <= 2^14) and generic ----------------
def test_generic_kernels_agree_with_fused(bfv, ckks, oracle_mod, monkeypatch):
    o, g = bfv
    a, b = _enc(o, oracle_mod, D1, 1), _enc(o, oracle_mod, D2, 2)
    oc, gc, x, y, cx, cy, scale = ckks
    monkeypatch.setenv("ABC_HIP_NO_FUSED", "1")
    m3 = o.multiply(a, b)
    _report("generic bfv relinearize", g.relinearize(m3), o.relinearize(m3))
    _report("generic bfv rotate", g.rotate(a, 5), o.rotate(a, 5))
    _report("generic ckks mul_relin", gc.mul_relin(cx, cy), oc.mul_relin(cx, cy))
    _report("generic ckks rotate", gc.rotate(cx, -7), oc.rotate(cx, -7))
    monkeypatch.delenv("ABC_HIP_NO_FUSED")
    _report("fused bfv relinearize", g.relinearize(m3), o.relinearize(m3))
    _report("fused bfv rotate", g.rotate(a, 5), o.rotate(a, 5))
    _report("fused ckks rotate", gc.rotate(cx, -7), oc.rotate(cx, -7))
    rs = oc.rescale(oc.mul_relin(cx, cy))
    _report("fused ckks rotate at level 3", gc.rotate(rs, 64), oc.rotate(rs, 64))
    ks = oc.keyswitch(cx[1], oc.relin_key())
    _report("fused raw keyswitch", gc.keyswitch(cx[1], 0), ks)
