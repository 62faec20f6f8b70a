"""CPU model of the fp64 residue arithmetic of abc_amd/csrc/abc_ntt.hpp ("fp64 residue arithmetic", FpArith).

The HIP path claims: every step of the fp64 butterfly is EXACT as long as magnitudes stay below 2^53, so the
transform returns the same residues as the integer algorithm; 49/50-bit primes must re-centre at every register pass,
smaller primes never inside a forward transform.  This test replays the device algorithm with Python integers for the
exact parts and IEEE doubles for the two inexact ones (the stored w/q and the quotient estimate), on adversarial inputs
(all q-1, alternating 0 / q-1, (q +- 1)/2), and checks
  * the FMA steps really are exact (h - c q and the low product part fit a double without rounding),
  * no magnitude reaches 2^53 under the pass schedule the kernels use (4,4,4,2 at N = 2^14),
  * the canonicalised output equals the oracle's NTT (tests/test_oracle_golden.py pins the oracle),
  * and that the re-centring the schedule prescribes is NECESSARY: without it a 50-bit prime does leave the exact range.
No GPU, no HIP library: this is the host-side proof obligation of the device code.
"""
import numpy as np
import pytest

LIMIT = 1 << 53


def _pow(a, e, q):
    return pow(int(a), int(e), int(q))


def _min_root(two_n, q):
    cof = (q - 1) // two_n
    g = next(c for c in (_pow(x, cof, q) for x in range(2, 1000)) if _pow(c, two_n // 2, q) == q - 1)
    sq, best, cur = g * g % q, g, g
    for _ in range(two_n // 2):
        best = min(best, cur)
        cur = cur * sq % q
    return best


def _bitrev(x, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


def _tables(n, q):
    logn = n.bit_length() - 1
    psi = _min_root(2 * n, q)
    tw = [0] * n
    p = 1
    for i in range(n):
        tw[_bitrev(i, logn)] = p
        p = p * psi % q
    # device table: {w centred, fl(w / q)}
    qd = float(q)
    out = []
    for w in tw:
        wc = w - q if w > q // 2 else w
        out.append((wc, float(wc) / qd))
    return out


class Model:
    """exact replay of fp_mul_lazy / fp_centre with range bookkeeping"""

    def __init__(self, q):
        self.q, self.qd, self.qinv = q, float(q), 1.0 / float(q)
        self.peak = 0

    def _see(self, v):
        a = abs(v)
        if a > self.peak:
            self.peak = a

    def mul_lazy(self, y, wc, wq):
        # h = fl(y w), l = y w - h exactly (FMA); both must be representable
        prod = y * wc
        h = float(prod)                      # correctly rounded int -> double
        l = prod - int(h)
        assert float(l) == l, "low product part is not a double"
        c = int(np.rint(np.float64(y) * np.float64(wq)))   # the only approximate quantity
        d = int(h) - c * self.q
        assert abs(d) < LIMIT and float(d) == d, "h - c q left the exact range"
        v = d + l
        self._see(v)
        assert abs(v) < LIMIT
        return v

    def centre(self, x):
        c = int(np.rint(np.float64(x) * np.float64(self.qinv)))
        r = x - c * self.q
        assert abs(r) <= self.q // 2 + 1
        return r


def _fp_forward(x, q, tables, schedule, recentre):
    """Cooley-Tukey, natural in -> bit-reversed out, stages grouped into register passes like ntt_fwd_block_a.
    recentre: True (before every pass but the first: FpArith for 49/50-bit primes), False, or the set of passes that start with
    a re-centring (FpTail: {2} for 50-bit primes, nothing for smaller ones)"""
    n = len(x)
    m = Model(q)
    x = list(x)
    stage = 0
    for pno, r in enumerate(schedule):
        if (pno in recentre) if isinstance(recentre, (set, frozenset)) else (pno > 0 and recentre):
            x = [m.centre(v) for v in x]
        for _ in range(r):
            half = n >> (stage + 1)
            blocks = 1 << stage
            for b in range(blocks):
                wc, wq = tables[blocks + b]
                base = b * 2 * half
                for j in range(base, base + half):
                    a = x[j]
                    v = m.mul_lazy(x[j + half], wc, wq)
                    x[j], x[j + half] = a + v, a - v
                    m._see(x[j]); m._see(x[j + half])
            stage += 1
    assert m.peak < LIMIT
    return [v % q for v in x], m.peak


CASES = [(50, True), (40, False)]


@pytest.mark.parametrize("bits,red", CASES)
def test_fp64_forward_transform_is_exact_and_matches_oracle(bits, red):
    from oracle import oracle_py as om
    n, logn = 4096, 12  # the per-stage growth argument does not depend on N; 2^12 keeps pure Python in seconds
    primes = om.create_primes(n, [bits, 40 if bits == 50 else 41])
    q = primes[0]
    o = om.Oracle(om.CKKS, n, primes)
    tables = _tables(n, q)
    rng = np.random.default_rng(3)
    inputs = {
        "all q-1": [q - 1] * n,
        "alternating": [(q - 1) if i & 1 else 0 for i in range(n)],
        "half": [((q - 1) // 2) if i % 3 else ((q + 1) // 2) for i in range(n)],
        "random": [int(v) for v in rng.integers(0, q, size=n, dtype=np.uint64)],
    }
    for name, x in inputs.items():
        got, peak = _fp_forward(x, q, tables, (4, 4, 2, 2), recentre=red)
        want = o.ntt(0, np.array(x, dtype=np.uint64))
        assert got == [int(v) for v in want], name
        # stated bounds: 8q for the re-centring primes, 2^52 for the others
        assert peak < (8 * q if red else 1 << 52), (name, peak / q)


@pytest.mark.parametrize("bits,schedule,recentre", [
    (50, (4, 4, 2), {2}), (50, (3, 4, 3), {2}), (49, (4, 4, 2), set()), (49, (3, 4, 3), set()),
    (50, (4, 4, 2, 2), {2}), (49, (4, 4, 2, 2), set()),
])
def test_fp64_block_tail_policy_from_centred_inputs(bits, schedule, recentre):
    """FpTail (abc_ntt.hpp): a forward block tail of ten (1024 points) or twelve (4096 points) stages that starts from CENTRED
    values re-centres once for a 50-bit prime (before pass 2) and never for a 49-bit one -- every step exact, every magnitude
    below 2^53, residues equal to the oracle's transform."""
    from oracle import oracle_py as om
    n = 1 << sum(schedule)
    primes = om.create_primes(n, [bits, 40])
    q = primes[0]
    o = om.Oracle(om.CKKS, n, primes)
    tables = _tables(n, q)
    rng = np.random.default_rng(5)
    half = (q - 1) // 2
    inputs = {  # centred representatives, |x| <= q/2
        "all +q/2": [half] * n,
        "all -q/2": [-half] * n,
        "alternating": [half if i & 1 else -half for i in range(n)],
        "blocks": [half if (i >> 3) & 1 else -half for i in range(n)],
        "random": [int(v) - half for v in rng.integers(0, q, size=n, dtype=np.uint64)],
    }
    for name, x in inputs.items():
        got, peak = _fp_forward(x, q, tables, schedule, recentre)
        want = o.ntt(0, np.array([v % q for v in x], dtype=np.uint64))
        assert got == [int(v) for v in want], name
        assert peak < LIMIT, (name, peak / q)
    # the bound the policy rests on: eight stages of a 50-bit prime from q/2 stay below 2^53, a ninth would not;
    # twelve stages of a 49-bit prime stay below it with room to spare
    def grow(y, qq, stages):
        for _ in range(stages):
            y = y * (1.0 + qq * 2.0 ** -53) + qq / 2.0
        return y
    q50 = float((1 << 50) - 1)
    assert grow(q50 / 2, q50, 8) < 2.0 ** 53 < grow(q50 / 2, q50, 9)
    q49 = float((1 << 49) - 1)
    assert grow(q49 / 2, q49, 12) < 2.0 ** 52.5


def test_recentring_is_necessary_for_50_bit_primes():
    """the same 14-stage transform WITHOUT the per-pass re-centring leaves the exact range for a 50-bit prime"""
    from oracle import oracle_py as om
    n = 16384
    q = om.create_primes(n, [50, 40])[0]
    # growth recurrence of the magnitude bound, Y' = Y (1 + q 2^-53) + q/2, 14 stages from a canonical input
    y = float(q)
    for _ in range(14):
        y = y * (1.0 + q * 2.0 ** -53) + q / 2.0
    assert y > 2.0 ** 53
    # with re-centring at every pass (4 stages from |x| <= q/2, or from a canonical input in the first pass)
    worst = 0.0
    for start in (float(q), q / 2.0 + 1):
        y = start
        for _ in range(4):
            y = y * (1.0 + q * 2.0 ** -53) + q / 2.0
        worst = max(worst, y)
    assert worst < 8.0 * q < 2.0 ** 53
    # and a 48-bit prime needs none, even fed a 50-bit operand (key-switch decomposition)
    q48 = om.create_primes(n, [48, 40])[0]
    y = 2.0 ** 50
    for _ in range(14):
        y = y * (1.0 + q48 * 2.0 ** -53) + q48 / 2.0
    assert y < 2.0 ** 52


def test_fp64_inverse_butterfly_bounds():
    """Gentleman-Sande: X = a + b doubles per stage; four stages from |x| <= q/2 give (a - b) <= 8q at the last one"""
    for bits in (40, 48, 50):
        q = float((1 << bits) - 1)
        y = q / 2
        for _ in range(4):
            diff = 2 * y                          # |a - b|
            assert diff <= 8 * q and diff < 2.0 ** 53
            yv = q / 2 + diff * q * 2.0 ** -53    # |(a - b) w| after the lazy product
            y = max(2 * y, yv)                    # |a + b|
        assert y <= 8 * q


def test_packed_half_done_limb_encoding_round_trips():
    """abc_ntt.hpp, "packed half-done limbs": a centred residue c (|c| <= q/2) is stored as the low dword of bits(c + 1.5 * 2^52) plus the
    next byte (primes of at most 40 bits) or half-word (at most 48 bits); the reader rebuilds bits(...) from the SIGN-EXTENDED high part
    and subtracts the constant.  Replayed here with numpy's IEEE doubles on the ends of the range and random values."""
    magic = np.float64(6755399441055744.0)  # 2^52 + 2^51
    rng = np.random.default_rng(17)
    for bits, hi_bits in ((40, 8), (36, 8), (48, 16), (41, 16)):
        half = (1 << (bits - 1)) - 1
        vals = np.concatenate([np.array([0, 1, -1, half, -half, half - 1, -(half - 1), 255, -256, 1 << 32, -(1 << 32), (1 << 32) - 1],
                                        dtype=np.int64),
                               rng.integers(-half, half + 1, size=20000, dtype=np.int64)])
        c = vals.astype(np.float64)
        assert np.array_equal(c.astype(np.int64), vals)            # exactly representable
        b = (c + magic).view(np.uint64)
        lo = (b & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        hi = ((b >> np.uint64(32)) & np.uint64((1 << hi_bits) - 1)).astype(np.uint32)
        # reader: sign-extend the stored high part, add it to the high word of the constant's bit pattern
        hs = hi.astype(np.int64)
        hs = np.where(hs >= (1 << (hi_bits - 1)), hs - (1 << hi_bits), hs)
        top = (np.int64(0x43380000) + hs).astype(np.uint64)
        rebuilt = ((top << np.uint64(32)) | lo.astype(np.uint64)).view(np.float64) - magic
        assert np.array_equal(rebuilt, c), bits
