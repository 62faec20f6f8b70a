"""GPU parity of every kernel path the dispatcher can take for the N = 2^14 hot call.

The default CKKS N=2^14 path is: fp64 transforms (all key primes < 2^50), tensor product + inverse transform + register
pass in one kernel, single-wavefront tail transforms fused with the key inner product.  Each switch below turns one of
those choices off (the same switches `tools/` uses for A/B timing); all variants must return the oracle's residues bit
for bit -- on ordinary ciphertexts and on adversarial ones whose residues sit at the ends of [0, q), which is where the
fp64 path's magnitude bounds (abc_ntt.hpp, "fp64 residue arithmetic") would break first.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

VARIANTS = {
    "default": {},
    "integer_transforms": {"ABC_HIP_NO_FP64": "1"},  # integer split kernels (abc_kernels_isplit.hip), unguarded
    "integer_v1": {"ABC_HIP_NO_FP64": "1", "ABC_HIP_NO_ISPLIT": "1"},  # round-1 integer sequence (LDS-resident transforms)
    "fp64_unsplit": {"ABC_HIP_NO_SPLIT": "1"},  # LDS-resident kernels (the sequence every ring below 2^14 takes)
    "fp64_fat_front": {"ABC_HIP_NO_LEAN_FRONT": "1"},  # the 139 KiB tensor / operand kernel even for small batches
    "fp64_unpacked": {"ABC_HIP_NO_PACK": "1"},  # half-done limbs as raw doubles (default: 5 / 6 bytes for primes <= 40 / 48 bits)
    "fp64_unpacked_fat_front": {"ABC_HIP_NO_PACK": "1", "ABC_HIP_NO_LEAN_FRONT": "1"},
    "fp64_u64_keys": {"ABC_HIP_NO_KEY_TWIN": "1"},  # key words converted per use instead of read from the key's fp64 twin
    "fp64_split3_main": {"ABC_HIP_NO_SPLIT4": "1"},  # previous generation of the last step (also what six and seven data limbs take)
    "generic": {"ABC_HIP_NO_FUSED": "1"},
    "no_galois_fusion_sync_alloc": {"ABC_HIP_NO_GALOIS_FUSION": "1", "ABC_HIP_SYNC_ALLOC": "1"},
}


def _same(name, got, want):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, name
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)
        raise AssertionError("%s: %d/%d words differ, first at %s: got %d want %d" % (
            name, len(bad), got.size, tuple(bad[0]), got[tuple(bad[0])], want[tuple(bad[0])]))


def _extreme_ct(primes, nl, n, rng):
    """a 2-component 'ciphertext' whose residues are drawn from {0, 1, (q-1)/2, (q+1)/2, q-2, q-1} and random values"""
    ct = np.empty((2, nl, n), dtype=np.uint64)
    for j in range(nl):
        q = primes[j]
        pool = np.array([0, 1, (q - 1) // 2, (q + 1) // 2, q - 2, q - 1], dtype=np.uint64)
        pick = rng.integers(0, 8, size=(2, n))
        rnd = rng.integers(0, q, size=(2, n), dtype=np.uint64)
        ct[:, j, :] = np.where(pick < 6, pool[np.minimum(pick, 5)], rnd)
    ct[0, :, : n // 4] = np.array(primes[:nl], dtype=np.uint64)[:, None] - 1  # long runs of q-1
    return ct


@pytest.fixture(scope="module")
def oracle14(oracle_mod):
    n = 16384
    primes = oracle_mod.create_primes(n, [50, 40, 40, 40, 50])
    o = oracle_mod.Oracle(oracle_mod.CKKS, n, primes)
    o.keygen(0xABC00001)
    rng = np.random.default_rng(2026)
    x, y = rng.uniform(-1, 1, n // 2), rng.uniform(-1, 1, n // 2)
    cx, cy = o.encrypt(o.ckks_encode(x, 2.0 ** 40), 11), o.encrypt(o.ckks_encode(y, 2.0 ** 40), 12)
    ex, ey = _extreme_ct(primes, 4, n, rng), _extreme_ct(primes, 4, n, rng)
    want = {
        "mul": o.mul_relin(cx, cy),
        "mul_extreme": o.mul_relin(ex, ey),
        "rot": o.rotate(cx, -7),
        "rot_extreme": o.rotate(ex, 3),
    }
    lvl3 = o.rescale(want["mul"])
    want["rot_l3"] = o.rotate(lvl3, 64)
    return o, primes, dict(cx=cx, cy=cy, ex=ex, ey=ey, lvl3=lvl3), want


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_ckks14_paths_bit_exact(variant, oracle14, capi, monkeypatch):
    o, primes, ins, want = oracle14
    for k, v in VARIANTS[variant].items():
        monkeypatch.setenv(k, v)
    g = capi.Context(capi.CKKS, o.n, primes)  # ABC_HIP_NO_FP64 is read when the context is built
    g.load_keys(sk=o.secret_key(), pk=o.public_key(), relin=o.relin_key(),
                galois={e: o.galois_key(e) for e in o.galois_elts()})
    _same(variant + " mul_relin", g.mul_relin(ins["cx"], ins["cy"]), want["mul"])
    _same(variant + " mul_relin extreme residues", g.mul_relin(ins["ex"], ins["ey"]), want["mul_extreme"])
    _same(variant + " rotate", g.rotate(ins["cx"], -7), want["rot"])
    _same(variant + " rotate extreme residues", g.rotate(ins["ex"], 3), want["rot_extreme"])
    _same(variant + " rotate at level 3", g.rotate(ins["lvl3"], 64), want["rot_l3"])
    # batched call: 5 pairs in one launch, mixed ordinary / extreme
    a = np.stack([ins["cx"], ins["ex"], ins["cy"], ins["ey"], ins["cx"]])
    b = np.stack([ins["cy"], ins["ey"], ins["cx"], ins["ex"], ins["cx"]])
    got = g.mul_relin(a, b)
    _same(variant + " batch[0]", got[0], want["mul"])
    _same(variant + " batch[1]", got[1], want["mul_extreme"])
    _same(variant + " batch[2]", got[2], o.mul_relin(ins["cy"], ins["cx"]))
    _same(variant + " batch[3]", got[3], o.mul_relin(ins["ey"], ins["ex"]))
    g.close()


@pytest.mark.parametrize("variant", ["default", "fp64_fat_front", "fp64_unpacked", "integer_transforms", "integer_v1", "fp64_unsplit", "fp64_split3_main"])
def test_ckks14_every_level_bit_exact(variant, oracle14, capi, monkeypatch):
    """multiply + relinearise and rotate at data levels 4, 3, 2 and 1 (the cooperative tail kernel runs nl wavefronts)"""
    o, primes, ins, want = oracle14
    for k, v in VARIANTS[variant].items():
        monkeypatch.setenv(k, v)
    g = capi.Context(capi.CKKS, o.n, primes)
    g.load_keys(sk=o.secret_key(), pk=o.public_key(), relin=o.relin_key(),
                galois={e: o.galois_key(e) for e in o.galois_elts()})
    x, y = ins["cx"], ins["cy"]
    for level in (4, 3, 2, 1):
        assert x.shape[1] == level
        _same("%s mul_relin at level %d" % (variant, level), g.mul_relin(x, y), o.mul_relin(x, y))
        _same("%s rotate at level %d" % (variant, level), g.rotate(x, 5), o.rotate(x, 5))
        if level > 1:
            _same("%s rescale at level %d" % (variant, level), g.rescale(x), o.rescale(x))
            both = np.stack([x, y, x])  # batched, odd count: exercises the ragged XCD group
            _same("%s batched rescale at level %d" % (variant, level), g.rescale(both)[1], o.rescale(y))
        if level > 1:
            x, y = o.mod_switch(x), o.mod_switch(y)
    g.close()


VARIANTS["bfv_v1_keyswitch"] = {"ABC_HIP_NO_BSPLIT": "1"}  # LDS-atomic tail kernel + LDS-resident mod-down (round 1)
VARIANTS["bfv_unsplit_multiply"] = {"ABC_HIP_NO_BMUL": "1"}  # LDS-resident BEHZ multiply (six kernels) + the split key switch
VARIANTS["bfv_special_one_round"] = {"ABC_HIP_NO_SPECIAL8X2": "1"}  # key-switch inner product: eight wavefronts, one round
VARIANTS["bfv_separate_permutation"] = {"ABC_HIP_NO_GALOIS_FUSION": "1"}  # rotations: k_galois first, then the key switch
VARIANTS["bfv_seal_aux_base"] = {"ABC_HIP_BEHZ_SEAL_BASE": "1"}  # 61-bit BEHZ auxiliary primes, as SEAL draws them
VARIANTS["bfv_int_behz_kernels"] = {"ABC_HIP_BEHZ_INT_KERNELS": "1"}  # 50-bit base, integer base-conversion kernels


@pytest.mark.parametrize("variant", ["default", "bfv_unsplit_multiply", "bfv_special_one_round", "bfv_v1_keyswitch", "bfv_separate_permutation", "bfv_seal_aux_base", "bfv_int_behz_kernels", "integer_transforms", "fp64_unsplit"])
def test_bfv14_keyswitch_paths_bit_exact(variant, oracle_mod, capi, monkeypatch):
    """BFVDefault(16384): 48/49-bit primes, i.e. the re-centring ('red') fp64 butterflies, coefficient-form operand"""
    for k, v in VARIANTS[variant].items():
        monkeypatch.setenv(k, v)
    o = oracle_mod.Oracle.bfv_default(16384)
    o.keygen(0xABC00003)
    g = capi.Context.bfv_default(16384)
    g.load_keys(sk=o.secret_key(), pk=o.public_key(), relin=o.relin_key(),
                galois={e: o.galois_key(e) for e in o.galois_elts()})
    rng = np.random.default_rng(5)
    nl = len(o.primes) - 1
    ex = _extreme_ct(o.primes, nl, o.n, rng)
    _same(variant + " bfv rotate extreme", g.rotate(ex, 1), o.rotate(ex, 1))
    ct = o.encrypt(o.encode(oracle_mod.expand_vector([3, 1, 4, 1, 5], o.n)), 9)
    _same(variant + " bfv mul_relin", g.mul_relin(ct, ct), o.mul_relin(ct, ct))
    # multiply_plain: one call (spread-out transforms) and a batch of four (fused forward / product / inverse kernel)
    pl = o.encode(oracle_mod.expand_vector([7, 0, 2, 5], o.n))
    _same(variant + " bfv multiply_plain", g.multiply_plain(ct, pl), o.multiply_plain(ct, pl))
    batch = np.stack([ct, ex, ct, ex])
    got = g.multiply_plain(batch, pl)
    _same(variant + " bfv multiply_plain batch[1]", got[1], o.multiply_plain(ex, pl))
    _same(variant + " bfv multiply_plain batch[2]", got[2], o.multiply_plain(ct, pl))
    # batched key switches: 40 ciphertexts (more than 128 (ciphertext, limb) pairs: the one-workgroup-per-limb register pass),
    # mixed ordinary / adversarial, rotation by a step that needs the NAF decomposition
    big = np.stack([ct if i % 3 else ex for i in range(40)])
    rot = g.rotate(big, -5)
    _same(variant + " bfv batched rotate [0]", rot[0], o.rotate(ex, -5))
    _same(variant + " bfv batched rotate [39]", rot[39], o.rotate(ex, -5))
    _same(variant + " bfv batched rotate [1]", rot[1], o.rotate(ct, -5))
    mr = g.mul_relin(big, big[::-1].copy())
    _same(variant + " bfv batched mul_relin [1]", mr[1], o.mul_relin(ct, ct))
    _same(variant + " bfv batched mul_relin [0]", mr[0], o.mul_relin(ex, ex))
    g.close()


@pytest.mark.parametrize("variant", ["default", "bfv_unsplit_multiply", "bfv_v1_keyswitch", "bfv_separate_permutation", "bfv_seal_aux_base", "integer_transforms"])
def test_bfv13_split_paths_bit_exact(variant, oracle_mod, capi, monkeypatch):
    """BFVDefault(8192) (four 43/44-bit data primes + special): the split multiply (k_bmul_front / _mid / _back<13, 3, 4, 4>, radix-8
    cross passes over eight 1024-point blocks) and the split key switch (k_bsplit_pass0 / k_gsplit_special / k_bsplit_tcoef /
    k_bsplit_finish_big<13>) against the oracle, and the LDS-resident kernels they replace (ABC_HIP_NO_BMUL / ABC_HIP_NO_BSPLIT)."""
    for k, v in VARIANTS[variant].items():
        monkeypatch.setenv(k, v)
    o = oracle_mod.Oracle.bfv_default(8192)
    o.keygen(0xABC00013)
    g = capi.Context.bfv_default(8192)
    g.load_keys(sk=o.secret_key(), pk=o.public_key(), relin=o.relin_key(),
                galois={e: o.galois_key(e) for e in o.galois_elts()})
    rng = np.random.default_rng(13)
    nl = len(o.primes) - 1
    ex = _extreme_ct(o.primes, nl, o.n, rng)
    ct = o.encrypt(o.encode(oracle_mod.expand_vector([3, 1, 4, 1, 5], o.n)), 9)
    _same(variant + " bfv13 multiply (size 3)", g.multiply(ct, ex), o.multiply(ct, ex))
    _same(variant + " bfv13 mul_relin", g.mul_relin(ct, ct), o.mul_relin(ct, ct))
    _same(variant + " bfv13 mul_relin extreme", g.mul_relin(ex, ex), o.mul_relin(ex, ex))
    _same(variant + " bfv13 rotate extreme", g.rotate(ex, 1), o.rotate(ex, 1))
    t3 = o.multiply(ct, ct)
    _same(variant + " bfv13 relinearize", g.relinearize(t3), o.relinearize(t3))
    big = np.stack([ct if i % 3 else ex for i in range(37)])  # ragged against every group size in the kernels
    rot = g.rotate(big, -5)
    _same(variant + " bfv13 batched rotate [0]", rot[0], o.rotate(ex, -5))
    _same(variant + " bfv13 batched rotate [36]", rot[36], o.rotate(ex, -5))
    _same(variant + " bfv13 batched rotate [1]", rot[1], o.rotate(ct, -5))
    mr = g.mul_relin(big, big[::-1].copy())
    _same(variant + " bfv13 batched mul_relin [1]", mr[1], o.mul_relin(ct, ct))
    _same(variant + " bfv13 batched mul_relin [0]", mr[0], o.mul_relin(ex, ex))
    _same(variant + " bfv13 batched mul_relin [35]", mr[35], o.mul_relin(ct, ct))
    g.close()


def test_fp64_and_integer_paths_agree_on_random_residues(oracle14, capi, monkeypatch):
    """2 000 random ciphertext pairs (uniform residues, plus rows forced to the ends of [0, q)): the fp64 kernels and the
    integer kernels must agree word for word -- a rounding slip in the fp64 quotient estimate that left the exact range
    would show up here long before it showed up in a decrypted slot."""
    o, primes, _, _ = oracle14
    keys = dict(sk=o.secret_key(), pk=o.public_key(), relin=o.relin_key(),
                galois={e: o.galois_key(e) for e in o.galois_elts()})
    g_fp = capi.Context(capi.CKKS, o.n, primes)
    g_fp.load_keys(**keys)
    monkeypatch.setenv("ABC_HIP_NO_FP64", "1")
    g_int = capi.Context(capi.CKKS, o.n, primes)
    g_int.load_keys(**keys)
    monkeypatch.delenv("ABC_HIP_NO_FP64")
    rng = np.random.default_rng(99)
    B = 200
    for rnd in range(10):
        nl = int(rng.integers(1, 5))
        a = np.stack([rng.integers(0, q, size=(B, 2, o.n), dtype=np.uint64) for q in primes[:nl]], axis=2)
        b = np.stack([rng.integers(0, q, size=(B, 2, o.n), dtype=np.uint64) for q in primes[:nl]], axis=2)
        for j in range(nl):  # a few adversarial rows per batch
            a[0, :, j, :] = primes[j] - 1
            b[0, :, j, :] = primes[j] - 1
            a[1, :, j, ::2] = 0
            b[1, :, j, 1::2] = (primes[j] - 1) // 2
        _same("round %d mul_relin nl=%d" % (rnd, nl), g_fp.mul_relin(a, b), g_int.mul_relin(a, b))
        steps = int(rng.integers(1, o.n // 2))
        _same("round %d rotate %d nl=%d" % (rnd, steps, nl), g_fp.rotate(a, steps), g_int.rotate(a, steps))
        if nl > 1:
            _same("round %d rescale nl=%d" % (rnd, nl), g_fp.rescale(a), g_int.rescale(a))
    g_fp.close()
    g_int.close()


# ---------------------------------------------------------------------------------------------------------------------
# SEAL-typical chains with primes above 2^50: the integer kernels, guarded (58..60 bits) and unguarded (51..57 bits),
# with and without the lazy 128-bit inner product (<= 55 bits)
# ---------------------------------------------------------------------------------------------------------------------
WIDE_CHAINS = {
    "60_40_40_60": [60, 40, 40, 60],      # guarded butterflies (k_fused_ks_decomp_ntt<LB,true,false>, moddown<LB,true>)
    "57_45_45_57": [57, 45, 45, 57],      # unguarded, non-lazy inner product (<LB,false,false>, moddown<LB,false>)
    "55_52_51_55": [55, 52, 51, 55],      # unguarded + lazy (<LB,false,true>) with every prime above the fp64 limit
    "60_50_40_50": [60, 50, 40, 50],      # mixed: fp64-capable data primes next to a 60-bit one (those limbs take the fp64 kernels)
    "60_40_40_40_60": [60, 40, 40, 40, 60],  # bench.py's value_60bit_primes chain: integer ends, fp64 middle
    "50_40_58_40_50": [50, 40, 58, 40, 50],  # an integer prime between fp64 ones (the main kernels' prime maps are not contiguous)
    "60_40x4_60": [60, 40, 40, 40, 40, 60],  # five data limbs: the split kernels' 5..7-limb instantiations
    "60_45x6_60": [60, 45, 45, 45, 45, 45, 45, 60],  # seven data limbs, mixed
    "50_40x6_50": [50, 40, 40, 40, 40, 40, 40, 50],  # seven data limbs, every prime below 2^50 (fp64 sequence)
}


@pytest.mark.parametrize("isplit", ["mixed", "integer_only", "round1"])
@pytest.mark.parametrize("chain", list(WIDE_CHAINS))
def test_ckks14_wide_prime_chains_every_level(chain, isplit, oracle_mod, capi, monkeypatch):
    # mixed (default): data primes below 2^50 take the fp64 kernels inside the integer sequence; integer_only: the same sequence
    # with integers throughout; round1: the round-1 integer kernels, which stay selectable and tested
    if isplit == "integer_only":
        monkeypatch.setenv("ABC_HIP_NO_MIXED", "1")
    if isplit == "round1":
        monkeypatch.setenv("ABC_HIP_NO_ISPLIT", "1")
    n = 16384
    primes = oracle_mod.create_primes(n, WIDE_CHAINS[chain])
    o = oracle_mod.Oracle(oracle_mod.CKKS, n, primes)
    o.keygen(0xABC00060)
    g = capi.Context(capi.CKKS, n, primes)
    g.load_keys(sk=o.secret_key(), pk=o.public_key(), relin=o.relin_key(),
                galois={e: o.galois_key(e) for e in o.galois_elts()})
    rng = np.random.default_rng(60)
    L = len(primes) - 1
    x = np.stack([rng.integers(0, q, size=(2, n), dtype=np.uint64) for q in primes[:L]], axis=1)
    y = _extreme_ct(primes, L, n, rng)
    for level in range(L, 0, -1):
        assert x.shape[1] == level
        _same("%s mul_relin level %d" % (chain, level), g.mul_relin(x, y), o.mul_relin(x, y))
        _same("%s rotate level %d" % (chain, level), g.rotate(y, 3), o.rotate(y, 3))
        _same("%s rotate (NAF) level %d" % (chain, level), g.rotate(x, 7), o.rotate(x, 7))
        if level > 1:
            _same("%s rescale level %d" % (chain, level), g.rescale(y), o.rescale(y))
            both = np.stack([x, y, x])
            _same("%s batched mul_relin level %d" % (chain, level), g.mul_relin(both, both[::-1].copy())[1], o.mul_relin(y, y))
            x, y = o.mod_switch(x), o.mod_switch(y)
    g.close()


# ---------------------------------------------------------------------------------------------------------------------
# packed half-done limbs (abc_ntt.hpp): 5 bytes per coefficient modulo primes of at most 40 bits, 6 bytes up to 48 bits, raw
# doubles above; every kind in one chain, every level, small batches (block-wise front) and large ones (139 KiB front kernel)
# ---------------------------------------------------------------------------------------------------------------------
PACK_CHAINS = {
    "50_46x3_50": [50, 46, 46, 46, 50],        # kind 2 limbs between raw ones
    "48_40_44_36_49": [48, 40, 44, 36, 49],    # kinds 2, 1, 2, 1 and a raw special prime
    "41_40_48_49_50_47": [41, 40, 48, 49, 50, 47],  # five data limbs (the widest chain the packed sequence takes), kind-2 special prime
}


@pytest.mark.parametrize("front", ["lean", "fat"])
@pytest.mark.parametrize("chain", list(PACK_CHAINS))
def test_ckks14_packed_half_done_limbs(chain, front, oracle_mod, capi, monkeypatch):
    if front == "fat":
        monkeypatch.setenv("ABC_HIP_NO_LEAN_FRONT", "1")
    n = 16384
    primes = oracle_mod.create_primes(n, PACK_CHAINS[chain])
    o = oracle_mod.Oracle(oracle_mod.CKKS, n, primes)
    o.keygen(0xABC00077)
    g = capi.Context(capi.CKKS, n, primes)
    g.load_keys(sk=o.secret_key(), pk=o.public_key(), relin=o.relin_key(),
                galois={e: o.galois_key(e) for e in o.galois_elts()})
    rng = np.random.default_rng(77)
    L = len(primes) - 1
    x = np.stack([rng.integers(0, q, size=(2, n), dtype=np.uint64) for q in primes[:L]], axis=1)
    y = _extreme_ct(primes, L, n, rng)
    for level in range(L, 0, -1):
        _same("%s mul_relin level %d" % (chain, level), g.mul_relin(x, y), o.mul_relin(x, y))
        _same("%s mul_relin extreme level %d" % (chain, level), g.mul_relin(y, y), o.mul_relin(y, y))
        _same("%s rotate level %d" % (chain, level), g.rotate(y, 3), o.rotate(y, 3))
        _same("%s relinearize-style rotate (NAF) level %d" % (chain, level), g.rotate(x, 7), o.rotate(x, 7))
        both = np.stack([x, y, x, y, y])
        got = g.mul_relin(both, both[::-1].copy())
        _same("%s batched mul_relin [1] level %d" % (chain, level), got[1], o.mul_relin(y, y))
        _same("%s batched mul_relin [0] level %d" % (chain, level), got[0], o.mul_relin(x, y))
        if level > 1:
            x, y = o.mod_switch(x), o.mod_switch(y)
    g.close()


# ---------------------------------------------------------------------------------------------------------------------
# several chunks per lane: the hot call splits a batch into chunks that alternate over internal streams and reuse
# per-lane scratch; every pair of a batch that spans chunk boundaries is checked, also with `out` aliasing `a`
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("variant", ["default", "fp64_fat_front", "fp64_unpacked", "integer_transforms", "integer_v1", "fp64_unsplit", "fp64_split3_main"])
def test_multi_chunk_batches_every_pair(variant, oracle14, capi, monkeypatch):
    import ctypes as C
    o, primes, ins, want = oracle14
    extra = dict(VARIANTS.get(variant, {}))
    for k, v in extra.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("ABC_HIP_CHUNK", "3")
    monkeypatch.setenv("ABC_HIP_LANES", "2")
    g = capi.Context(capi.CKKS, o.n, primes)
    g.load_keys(sk=o.secret_key(), pk=o.public_key(), relin=o.relin_key(),
                galois={e: o.galois_key(e) for e in o.galois_elts()})
    rng = np.random.default_rng(11)
    B = 11  # chunks of 3, 3, 3, 2 over two lanes: two chunks per lane, a ragged tail
    pool = [ins["cx"], ins["cy"], ins["ex"], ins["ey"]]
    a = np.stack([pool[int(i)] for i in rng.integers(0, 4, B)])
    b = np.stack([pool[int(i)] for i in rng.integers(0, 4, B)])
    a[5] = np.stack([rng.integers(0, q, size=(2, o.n), dtype=np.uint64) for q in primes[:4]], axis=1)
    expect = np.stack([o.mul_relin(a[i], b[i]) for i in range(B)])
    got = g.mul_relin(a, b)
    for i in range(B):
        _same("%s chunked mul_relin pair %d" % (variant, i), got[i], expect[i])
    # out aliasing a (multiplyInplace), then out aliasing b
    for alias in (0, 1):
        da, db = g.upload(a), g.upload(b)
        dst = (da, db)[alias]
        g.op("mul_relin", da.ptr, db.ptr, dst.ptr, 4, C.c_size_t(B))
        res = g.download(dst, a.shape)
        for i in range(B):
            _same("%s chunked in-place (alias %d) pair %d" % (variant, alias, i), res[i], expect[i])
        da.free(); db.free()
    rexp = np.stack([o.rotate(a[i], -7) for i in range(B)])
    rgot = g.rotate(a, -7)
    for i in range(B):
        _same("%s chunked rotate pair %d" % (variant, i), rgot[i], rexp[i])
    g.close()


def test_allocator_cache_trim_and_cap(capi):
    """alternating sizes: freed blocks are cached, abc_hip_trim returns them, a context destroyed with live buffers frees them"""
    n = 4096
    g = capi.Context.bfv_default(n)
    sizes = [1 << 20, 3 << 20]
    for rnd in range(6):
        bufs = [g.alloc(sizes[(rnd + k) % 2]) for k in range(4)]
        for b_ in bufs:
            b_.free()
    assert g.cached_bytes() == 2 * (1 << 20) + 2 * (3 << 20)  # two blocks per size were ever live at once
    g.trim()
    assert g.cached_bytes() == 0
    keep = g.alloc(1 << 20)  # still live at destroy: freed by the context
    g.close()
    keep.ptr = None
