"""Size-independent properties at BASELINE.json's full ring sizes, through the C ABI only (the device path end to
end: keygen, encode, encrypt, evaluate, decrypt, decode) -- no oracle involved, so these also run where the oracle
would take too long.  Edge cases follow the reference's tests: single value padded with itself, exactly N values,
more than N values rejected, negative values, empty batch."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bfv(capi, n):
    if n <= 32768:
        g = capi.Context.bfv_default(n)
    else:
        g = capi.Context(capi.BFV, n, capi.create_primes(n, [55] * 8 + [56]), capi.plain_modulus_batching(n, 20))
    g.keygen(0xABC00001)
    return g


def _enc(g, vals, seed):
    v = np.asarray(vals, dtype=np.int64).reshape(-1, g.n)
    return g.encrypt(g.batch_encode(v), seed)


def _dec(g, ct):
    return g.batch_decode(g.decrypt(ct))


@pytest.mark.parametrize("n", [4096, 16384, 65536])
def test_bfv_roundtrip_linearity_rotation(capi, n):
    g = _bfv(capi, n)
    rng = np.random.default_rng(n)
    t = g.t
    a = rng.integers(-1000, 1000, size=(2, n))
    b = rng.integers(-1000, 1000, size=(2, n))
    ca, cb = _enc(g, a, 1), _enc(g, b, 100)
    assert np.array_equal(_dec(g, ca), a)                                  # encode -> encrypt -> decrypt -> decode
    assert np.array_equal(_dec(g, g.add(ca, cb)), a + b)                   # linearity
    assert np.array_equal(_dec(g, g.sub(ca, cb)), a - b)
    assert np.array_equal(_dec(g, g.negate(ca)), -a)
    want = (a * b) % t
    want = np.where(want > t // 2, want - t, want)
    assert np.array_equal(_dec(g, g.mul_relin(ca, cb)), want)              # multiply then relinearize
    r = g.rotate(ca, 5)
    assert np.array_equal(_dec(g, g.rotate(r, -5)), a)                     # rotate k then -k is the identity
    row = n // 2
    d = _dec(g, r)
    assert np.array_equal(d[:, : row - 5], a[:, 5:row]) and np.array_equal(d[:, row - 5 : row], a[:, :5])  # row-wise wrap
    assert np.array_equal(d[:, row : n - 5], a[:, row + 5 :])
    one = g.batch_encode(np.ones((1, n), dtype=np.int64))
    assert np.array_equal(_dec(g, g.multiply_plain(ca, one)), a)           # times one
    zero = g.batch_encode(np.zeros((1, n), dtype=np.int64))
    assert np.array_equal(_dec(g, g.add_plain(ca, zero)), a)               # plus zero


def test_bfv_edge_cases(capi):
    n = 4096
    g = _bfv(capi, n)
    full = np.arange(n, dtype=np.int64) - n // 2                           # exactly N values, negatives included
    assert np.array_equal(_dec(g, _enc(g, full, 3))[0], full)
    single = np.full(n, 7, dtype=np.int64)                                 # a scalar is padded with itself (expandVector)
    assert np.array_equal(_dec(g, _enc(g, single, 4))[0], single)
    extreme = np.array([g.t // 2, -(g.t // 2)] * (n // 2), dtype=np.int64)  # largest representable magnitudes
    assert np.array_equal(_dec(g, _enc(g, extreme, 5))[0], extreme)
    # empty batch: every entry point accepts count = 0
    buf = g.alloc(8)
    g.op("add", buf.ptr, buf.ptr, buf.ptr, 2, g.L, C.c_size_t(0))
    g.op("mul_relin", buf.ptr, buf.ptr, buf.ptr, g.L, C.c_size_t(0))
    g.op("rotate", buf.ptr, buf.ptr, g.L, 1, C.c_size_t(0))
    g.sync()
    # wrong level / missing key / oversize step are errors, not silent fallbacks
    with pytest.raises(capi.AbcHipError):
        g.op("add", buf.ptr, buf.ptr, buf.ptr, 2, g.L + 1, C.c_size_t(1))
    with pytest.raises(capi.AbcHipError):
        g.op("rotate", buf.ptr, buf.ptr, g.L, n, C.c_size_t(1))
    g2 = capi.Context.bfv_default(n)
    with pytest.raises(capi.AbcHipError):
        g2.op("mul_relin", buf.ptr, buf.ptr, buf.ptr, g2.L, C.c_size_t(1))  # no relin key loaded


@pytest.mark.parametrize("n,bits", [(16384, [50, 40, 40, 40, 50]), (32768, [50, 40, 40, 50])])
def test_ckks_roundtrip_and_homomorphisms(capi, n, bits):
    from abc_amd import ckks_encoder as ce
    primes = capi.create_primes(n, bits)
    g = capi.Context(capi.CKKS, n, primes)
    g.keygen(9)
    L = g.L
    rng = np.random.default_rng(n)
    x, y = rng.uniform(-1, 1, n // 2), rng.uniform(-1, 1, n // 2)
    scale = 2.0 ** 40

    def enc(v, seed):
        return g.encrypt(g.ntt_limbs(ce.encode(v, scale, n, primes[:L])[None])[0], seed)

    def dec(ct, s):
        pl = g.ntt_limbs(g.decrypt(ct)[None], inverse=True)[0]
        return ce.decode(pl, s, n, primes).real

    cx, cy = enc(x, 1), enc(y, 2)
    assert np.abs(dec(cx, scale) - x).max() < 1e-6                 # tolerance: CKKS is approximate (2^40 scale)
    assert np.abs(dec(g.add(cx, cy), scale) - (x + y)).max() < 1e-6
    m = g.rescale(g.mul_relin(cx, cy))
    assert np.abs(dec(m, scale * scale / primes[L - 1]) - x * y).max() < 1e-5
    assert np.abs(dec(g.rotate(g.rotate(cx, 3), -3), scale) - x).max() < 1e-4
    assert np.abs(dec(g.rotate(cx, 1), scale) - np.roll(x, -1)).max() < 1e-4
