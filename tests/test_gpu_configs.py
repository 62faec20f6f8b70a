"""BASELINE.json configs 3-5 as circuits through the C ABI, checked against the CPU oracle (residues,
bit-exact) and against the cleartext computation (decrypted values).

  config 3  CKKS N=2^14, 4 limbs: dot product = mul+relin, rescale, rotate-and-add tree (the target form of the
            reference's Vectorizer, ref:test/visitor/VectorizerTest.cpp:169-173,209-214)
  config 4  CKKS N=2^15: 8x8 box sum on 64x64 images by separable log-tree rotations {1,2,4} then {64,128,256}
            (layout x*64+y and omitted normalisation as ref:test/end-to-end/BoxBlurTest.cpp:16-38)
  config 5  BFV N=2^16: depth-8 multiply chain (explicit primes; SEAL's default table stops at 2^15)
CKKS and the N >= 2^15 cases have no reference implementation or test: parity is GPU == oracle ("parity unpinned"
vs the reference).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _eq(name, got, want):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)
        raise AssertionError("%s: %d/%d words differ, first at %s" % (name, len(bad), got.size, tuple(bad[0])))


def _pair(oracle_mod, capi, scheme, n, primes, t=0, seed=0xABC00001):
    o = oracle_mod.Oracle(scheme, n, primes, t)
    o.keygen(seed)
    g = capi.Context(scheme, n, primes, t)
    g.keygen(seed)  # shared sampling spec: the device generates the identical keys
    return o, g


def test_config3_ckks_dot_product(oracle_mod, capi):
    n = 16384
    primes = oracle_mod.create_primes(n, [50, 40, 40, 40, 50])
    o, g = _pair(oracle_mod, capi, oracle_mod.CKKS, n, primes)
    rng = np.random.default_rng(3)
    x, y = rng.uniform(-1, 1, n // 2), rng.uniform(-1, 1, n // 2)
    scale = 2.0 ** 40
    cx, cy = o.encrypt(o.ckks_encode(x, scale), 1), o.encrypt(o.ckks_encode(y, scale), 2)

    def circuit(be):
        acc = be.rescale(be.mul_relin(cx, cy))
        step = n // 4  # 4096, 2048, ..., 1 : 13 rotations fold the 8192 slots
        while step >= 1:
            acc = be.add(acc, be.rotate(acc, step))
            step //= 2
        return acc

    want, got = circuit(o), circuit(g)
    _eq("dot product circuit", got, want)
    val = o.ckks_decode(o.decrypt(got), scale * scale / primes[3]).real
    assert abs(val[0] - float(np.dot(x, y))) < 1e-3  # CKKS tolerance: 8192 summands, 2^40 scale
    assert np.abs(val - val[0]).max() < 1e-3         # every slot holds the sum


def test_config4_ckks_box_blur_batch(oracle_mod, capi):
    n = 32768
    primes = oracle_mod.create_primes(n, [50, 40, 40, 50])
    o, g = _pair(oracle_mod, capi, oracle_mod.CKKS, n, primes)
    rng = np.random.default_rng(4)
    imgs = rng.integers(0, 1025, size=(2, 64, 64)).astype(np.float64)  # the reference's pixel range, BoxBlurTest.cpp:123
    scale = 2.0 ** 30
    cts = np.stack([o.encrypt(o.ckks_encode(im.reshape(-1), scale), 10 + i) for i, im in enumerate(imgs)])

    def circuit(be, ct):
        acc = ct
        for r in (1, 2, 4, 64, 128, 256):
            acc = be.add(acc, be.rotate(acc, r))
        return acc

    want = np.stack([circuit(o, c) for c in cts])
    got = circuit(g, cts)  # the whole batch in one call per op
    _eq("box blur batch", got, want)
    for b, im in enumerate(imgs):
        dec = o.ckks_decode(o.decrypt(got[b]), scale).real[:4096].reshape(64, 64)
        ref = sum(np.roll(np.roll(im, -dx, axis=0), -dy, axis=1) for dx in range(8) for dy in range(8))
        assert np.abs(dec[:56, :56] - ref[:56, :56]).max() < 1e-2  # interior: rotations do not wrap there


def test_config5_bfv_depth8_chain(oracle_mod, capi):
    n = 65536
    primes = oracle_mod.create_primes(n, [55] * 8 + [56])
    t = oracle_mod.plain_modulus_batching(n, 20)
    assert t == 786433
    o, g = _pair(oracle_mod, capi, oracle_mod.BFV, n, primes, t)
    vals = [[(3 * i + k) % 7 + 1 for i in range(16)] for k in range(9)]
    cts = [o.encrypt(o.encode(oracle_mod.expand_vector(v, n)), 50 + k) for k, v in enumerate(vals)]
    # residue parity on the first two levels (the oracle's BEHZ at this size takes seconds per multiply)
    r_o = o.mul_relin(cts[0], cts[1])
    r_g = g.mul_relin(cts[0], cts[1])
    _eq("N=2^16 mul_relin level 1", r_g, r_o)
    r_o2 = o.mul_relin(r_o, cts[2])
    r_g2 = g.mul_relin(r_g, cts[2])
    _eq("N=2^16 mul_relin level 2", r_g2, r_o2)
    acc = r_g2
    for k in range(3, 9):
        acc = g.mul_relin(acc, cts[k])
    want = [1] * 16
    for v in vals:
        want = [(a * b) % t for a, b in zip(want, v)]
    dec = o.decode(o.decrypt(acc))[:16] % t
    assert list(dec) == want
    assert o.noise_budget(acc) > 0


@pytest.mark.parametrize("n,bits", [(4096, [55, 55, 56]), (8192, [58, 52, 60, 60]), (16384, [60, 40, 40, 40, 60]), (32768, [55] * 8 + [56])])
@pytest.mark.parametrize("variant", ["default", "no_key_mirror", "unfused"])
def test_bfv_wide_chains(n, bits, variant, oracle_mod, capi, monkeypatch):
    """BFV on chains with primes above 2^50 (integer transforms, SEAL's 61-bit BEHZ auxiliary base): multiply, multiply + relinearise
    and a batch, on ordinary and on end-of-range residues (the base conversions' lazy sums at their largest).  N = 2^15: the key
    switch takes k_iks_pass0 / k_iks_special (abc_kernels_eval.hip) -- with the key's Shoup quotients, without them
    (ABC_HIP_NO_KEY_TWIN=1: Barrett products) -- unless ABC_HIP_NO_IKS=1."""
    if variant == "no_key_mirror":
        monkeypatch.setenv("ABC_HIP_NO_KEY_TWIN", "1")
    if variant == "unfused":
        monkeypatch.setenv("ABC_HIP_NO_IKS", "1")
    primes = oracle_mod.create_primes(n, bits)
    t = oracle_mod.plain_modulus_batching(n, 20)
    o, g = _pair(oracle_mod, capi, oracle_mod.BFV, n, primes, t, seed=91)
    a = o.encrypt(o.encode(oracle_mod.expand_vector([3, 1, 4, 1, 5, 9, 2, 6], n)), 1)
    b = o.encrypt(o.encode(oracle_mod.expand_vector([2, 7, 1, 8, 2, 8, 1, 8], n)), 2)
    rng = np.random.default_rng(n)
    ex = a.copy()  # end-of-range residues: the base conversions' lazy sums at their largest
    for i, q in enumerate(primes[:-1]):
        ex[:, i, :] = rng.choice(np.array([0, 1, q - 1, q - 2, q // 2, q // 2 + 1], dtype=np.uint64), size=(2, n))
    r = g.mul_relin(a, b)
    _eq("N=%d wide-chain mul_relin" % n, r, o.mul_relin(a, b))
    assert list(o.decode(o.decrypt(r))[:8]) == [6, 7, 4, 8, 10, 72, 2, 48]
    _eq("N=%d wide-chain multiply (3 components)" % n, g.multiply(r, b), o.multiply(r, b))
    _eq("N=%d wide-chain multiply, extreme residues" % n, g.multiply(ex, ex), o.multiply(ex, ex))
    got = g.mul_relin(np.stack([a, ex, r]), np.stack([b, a, ex]))
    _eq("N=%d wide-chain batch row 1" % n, got[1], o.mul_relin(ex, a))
    _eq("N=%d wide-chain batch row 2" % n, got[2], o.mul_relin(r, ex))
    # rotations: on the big ring the signed coefficient permutation is folded into k_iks_pass0 / k_iks_finish
    _eq("N=%d wide-chain rotate 1, extreme residues" % n, g.rotate(ex, 1), o.rotate(ex, 1))
    rot = g.rotate(np.stack([a, ex, r]), -7)  # needs the NAF decomposition: three key switches
    _eq("N=%d wide-chain batched rotate -7 [1]" % n, rot[1], o.rotate(ex, -7))
    _eq("N=%d wide-chain batched rotate -7 [2]" % n, rot[2], o.rotate(r, -7))


@pytest.mark.parametrize("generic", [False, True, "unfused_multiply", "multiply_on_4096_point_blocks"])
@pytest.mark.parametrize("n,bits", [(32768, [49] * 4 + [50]), (32768, [49] * 8 + [50]), (65536, [49] * 8 + [50])])
def test_big_ring_bfv_on_an_fp64_chain(n, bits, generic, oracle_mod, capi, monkeypatch):
    """BFV at N = 2^15 / 2^16 with every prime below 2^50 (config 5's alternative chain): the key switch takes the split
    kernels with a radix-32 / radix-64 cross pass (abc_kernels_gsplit.hip, k_bsplit_*); with ABC_HIP_NO_BSPLIT the generic
    sequence.  Eight data limbs (the BFVDefault shape): the multiply takes abc_kernels_bmul.hip's fused extension / floor kernels
    around the block tails (1024-point blocks behind two-level radix-32 / 64 cross passes; ABC_HIP_NO_BMUL_R6: 4096-point blocks);
    ABC_HIP_NO_BMUL selects the separate kernels.  All must give the oracle's residues."""
    if generic is True:
        monkeypatch.setenv("ABC_HIP_NO_BSPLIT", "1")
    if generic == "unfused_multiply":
        monkeypatch.setenv("ABC_HIP_NO_BMUL", "1")
    if generic == "multiply_on_4096_point_blocks":  # round 3's first form: radix-8 / 16 cross passes in registers around k_bmul_mid<.., 12>
        monkeypatch.setenv("ABC_HIP_NO_BMUL_R6", "1")
    primes = oracle_mod.create_primes(n, bits)
    t = oracle_mod.plain_modulus_batching(n, 20)
    o, g = _pair(oracle_mod, capi, oracle_mod.BFV, n, primes, t, seed=77)
    a = o.encrypt(o.encode(oracle_mod.expand_vector([3, 1, 4, 1, 5, 9, 2, 6], n)), 1)
    b = o.encrypt(o.encode(oracle_mod.expand_vector([2, 7, 1, 8, 2, 8, 1, 8], n)), 2)
    r = g.mul_relin(a, b)
    _eq("N=%d fp64-chain mul_relin" % n, r, o.mul_relin(a, b))
    assert list(o.decode(o.decrypt(r))[:8]) == [6, 7, 4, 8, 10, 72, 2, 48]
    _eq("N=%d fp64-chain rotate" % n, g.rotate(a, 5), o.rotate(a, 5))
    _eq("N=%d fp64-chain rotate -1" % n, g.rotate(r, -1), o.rotate(r, -1))
    # a batch of 3 (different rows) through the same path, in place on the first operand
    batch_a, batch_b = np.stack([a, b, r]), np.stack([b, r, a])
    want = np.stack([o.mul_relin(x, y) for x, y in zip(batch_a[:2], batch_b[:2])])
    got = g.mul_relin(batch_a, batch_b)
    _eq("N=%d fp64-chain batch rows 0-1" % n, got[:2], want)
    assert np.array_equal(got[2], g.mul_relin(r, a))


def test_bfv_default_8192_and_16384(oracle_mod, capi):
    """The reference factory's default ring is 16384 slots (SealCiphertextFactory.h:16): 8 data limbs + special."""
    for n in (8192, 16384):
        o, g = _pair(oracle_mod, capi, oracle_mod.BFV, n, oracle_mod.default_bfv_primes(n), oracle_mod.plain_modulus_batching(n, 20), seed=5)
        a = o.encrypt(o.encode(oracle_mod.expand_vector([3, 3, 1, 4, 5, 9], n)), 1)
        b = o.encrypt(o.encode(oracle_mod.expand_vector([0, 1, 2, 1, 10, 21], n)), 2)
        r = g.mul_relin(a, b)
        _eq("BFVDefault(%d) mul_relin" % n, r, o.mul_relin(a, b))
        assert list(o.decode(o.decrypt(r))[:6]) == [0, 3, 2, 4, 50, 189]
        _eq("BFVDefault(%d) rotate" % n, g.rotate(a, 3), o.rotate(a, 3))


def test_bfv_default_32768(oracle_mod, capi):
    """SEAL's BFVDefault(32768): fifteen 55-bit data primes + a special one (the largest ring of SEAL's parameter tables): integer
    kernels, 16 auxiliary primes, a key switch of 15 x 16 digit transforms through k_iks_pass0 / k_iks_special / k_iks_finish."""
    n = 32768
    o, g = _pair(oracle_mod, capi, oracle_mod.BFV, n, oracle_mod.default_bfv_primes(n), oracle_mod.plain_modulus_batching(n, 20), seed=6)
    assert g.L == 15
    a = o.encrypt(o.encode(oracle_mod.expand_vector([3, 3, 1, 4, 5, 9], n)), 1)
    b = o.encrypt(o.encode(oracle_mod.expand_vector([0, 1, 2, 1, 10, 21], n)), 2)
    r = g.mul_relin(a, b)
    _eq("BFVDefault(32768) mul_relin", r, o.mul_relin(a, b))
    assert list(o.decode(o.decrypt(r))[:6]) == [0, 3, 2, 4, 50, 189]
    _eq("BFVDefault(32768) rotate", g.rotate(r, -3), o.rotate(r, -3))
    got = g.mul_relin(np.stack([a, r]), np.stack([r, b]))
    _eq("BFVDefault(32768) batch row 1", got[1], o.mul_relin(r, b))


@pytest.mark.parametrize("variant", ["default", "no_key_mirror", "unfused"])
@pytest.mark.parametrize("n,bits", [(32768, [51, 57, 50, 50]), (32768, [60, 40, 40, 60]), (65536, [55, 45, 56])])
def test_ckks_big_ring_generic_sequence_with_the_fused_integer_key_switch(n, bits, variant, oracle_mod, capi, monkeypatch):
    """ABC_HIP_NO_FUSED=1 at N = 2^15 / 2^16 (N = 2^16 has no other CKKS sequence): generic kernels around k_iks_pass0 / k_iks_special,
    which keep the data limbs' sums in NTT form for CKKS and run in integers whatever the prime -- [51, 57, 50, 50] has an fp64-capable
    SPECIAL prime under integer data primes (the strided stages behind the kernel must then be the integer ones too: found by the
    randomised campaign) and takes the unguarded butterflies, [60, 40, 40, 60] the guarded ones.  Every level."""
    monkeypatch.setenv("ABC_HIP_NO_FUSED", "1")
    if variant == "no_key_mirror":
        monkeypatch.setenv("ABC_HIP_NO_KEY_TWIN", "1")
    if variant == "unfused":
        monkeypatch.setenv("ABC_HIP_NO_IKS", "1")
    primes = oracle_mod.create_primes(n, bits)
    o, g = _pair(oracle_mod, capi, oracle_mod.CKKS, n, primes, seed=0xABC00F16)
    rng = np.random.default_rng(n + len(bits))
    L = len(bits) - 1
    x = np.stack([rng.integers(0, q, size=(2, n), dtype=np.uint64) for q in primes[:L]], axis=1)
    y = np.stack([rng.integers(0, q, size=(2, n), dtype=np.uint64) for q in primes[:L]], axis=1)
    for j in range(L):
        y[0, j, : n // 4] = primes[j] - 1
        y[1, j, ::3] = 0
    for level in range(L, 0, -1):
        _eq("N=%d generic mul_relin level %d" % (n, level), g.mul_relin(x, y), o.mul_relin(x, y))
        _eq("N=%d generic rotate -7 level %d" % (n, level), g.rotate(y, -7), o.rotate(y, -7))
        if level == L:
            got = g.rotate(np.stack([x, y, x]), 3)
            _eq("N=%d generic batched rotate [1]" % n, got[1], o.rotate(y, 3))
        if level > 1:
            x, y = x[:, : level - 1].copy(), y[:, : level - 1].copy()


@pytest.mark.parametrize("bits", [[50, 40, 40, 40, 50], [60, 40, 40, 40, 60], [57, 45, 45, 57], [50, 58, 40, 55],
                                  [50, 40, 40, 40, 40, 40, 50], [60, 45, 45, 45, 45, 45, 45, 60],
                                  [50] + [40] * 8 + [50],   # nine data limbs: the deep-chain main kernel (levels 9 and 8)
                                  [60] + [40] * 8 + [60]])  # the same depth on a SEAL-typical chain: integer + fp64 deep kernels
@pytest.mark.parametrize("generic", [False, True, "integer_only"])
def test_ckks15_every_level_split_and_generic(generic, bits, oracle_mod, capi, monkeypatch):
    """N = 2^15: the split key switch without LDS-resident limbs (abc_kernels_gsplit.hip) and the generic kernels it
    replaces (ABC_HIP_NO_GSPLIT=1) against the oracle: multiply + relinearise, rotation (direct and NAF), relinearize and
    key switch stand-alone, at every level, single and batched (ragged group of four in the block-tail kernel)."""
    if generic == "integer_only":  # wide chains: no fp64 kernels for the limbs of primes below 2^50
        monkeypatch.setenv("ABC_HIP_NO_MIXED", "1")
    elif generic:
        monkeypatch.setenv("ABC_HIP_NO_GSPLIT", "1")
    n = 32768
    # every prime below 2^50: the fp64 sequence; a 60-bit (guarded) or 57-bit (unguarded) prime: its integer twin with 32 blocks
    primes = oracle_mod.create_primes(n, bits)
    o, g = _pair(oracle_mod, capi, oracle_mod.CKKS, n, primes, seed=0xABC00F15)
    rng = np.random.default_rng(15)
    L = len(bits) - 1
    x = np.stack([rng.integers(0, q, size=(2, n), dtype=np.uint64) for q in primes[:L]], axis=1)
    y = np.stack([rng.integers(0, q, size=(2, n), dtype=np.uint64) for q in primes[:L]], axis=1)
    for j in range(L):  # adversarial residues: long runs at the ends of [0, q)
        y[0, j, : n // 4] = primes[j] - 1
        y[1, j, ::3] = 0
    for level in range(L, 0, -1):
        assert x.shape[1] == level
        _eq("N=2^15 mul_relin level %d" % level, g.mul_relin(x, y), o.mul_relin(x, y))
        _eq("N=2^15 rotate 1 level %d" % level, g.rotate(y, 1), o.rotate(y, 1))
        _eq("N=2^15 rotate 5 (NAF) level %d" % level, g.rotate(x, 5), o.rotate(x, 5))
        if level == L:
            batch = np.stack([x, y, y, x, x])  # five pairs: one full group of four and a ragged one
            got = g.mul_relin(batch, batch[::-1].copy())
            _eq("N=2^15 batched mul_relin [1]", got[1], o.mul_relin(y, x))
            _eq("N=2^15 batched mul_relin [4]", got[4], o.mul_relin(x, x))
            _eq("N=2^15 batched rotate [2]", g.rotate(batch, -64)[2], o.rotate(y, -64))
            t3 = o.multiply(x, y)
            _eq("N=2^15 relinearize", g.relinearize(t3), o.relinearize(t3))
        if level > 1:
            x, y = o.mod_switch(x), o.mod_switch(y)
    g.close()
