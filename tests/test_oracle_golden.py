"""P0 parity: the CPU oracle reproduces every decrypted-slot golden vector the
reference's own SEAL-backed tests hold.  This pins the oracle; the GPU tests then pin the HIP kernels
to the oracle at residue level.

Vectors: test/runtime/SealCiphertextFactoryTest.cpp and test/runtime/RuntimeVisitorTest.cpp of the
reference (N = 4096, BFVDefault(4096), t = PlainModulus::Batching(4096, 20) = 1032193).
"""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def ctx(oracle_mod):
    o = oracle_mod.Oracle.bfv_default(4096)
    o.keygen(20261004)
    return o


def enc(o, om, vals, seed):
    return o.encrypt(o.encode(om.expand_vector(vals, o.n)), seed)


def dec(o, ct):
    return o.decode(o.decrypt(ct))


def check_padded(result, expected):
    """checkCiphertextData, SealCiphertextFactoryTest.cpp:22-41: values then last value repeated."""
    assert len(result) == 4096
    assert list(result[:len(expected)]) == list(expected)
    assert np.all(result[len(expected):] == expected[-1])


D1, D2 = [3, 3, 1, 4, 5, 9], [0, 1, 2, 1, 10, 21]
ROT = [123456, 3, 1, 4, 5, 9, 5, 2, 1, 5]


def test_parameters_match_seal_defaults(oracle_mod):
    # SealCiphertextFactory.cpp:80,83; SURVEY.md section 8: 36+36 | 37 bits, t = 1032193
    assert oracle_mod.default_bfv_primes(4096) == [0xffffee001, 0xffffc4001, 0x1ffffe0001]
    assert oracle_mod.plain_modulus_batching(4096, 20) == 1032193
    assert oracle_mod.plain_modulus_batching(8192, 20) == 1032193
    assert oracle_mod.plain_modulus_batching(16384, 20) == 786433
    for n in (1024, 2048, 4096, 8192, 16384, 32768):
        for p in oracle_mod.default_bfv_primes(n):
            assert oracle_mod.lib().orc_is_prime(p) and p % (2 * n) == 1


def test_create_ciphertext(ctx, oracle_mod):  # :44-49
    check_padded(dec(ctx, enc(ctx, oracle_mod, D1, 1)), D1)


def test_add_sub_multiply(ctx, oracle_mod):  # :146-192 and in-place twins :198-232
    a, b = enc(ctx, oracle_mod, D1, 1), enc(ctx, oracle_mod, D2, 2)
    check_padded(dec(ctx, ctx.add(a, b)), [3, 4, 3, 5, 15, 30])
    check_padded(dec(ctx, ctx.sub(a, b)), [3, 2, -1, 3, -5, -12])
    check_padded(dec(ctx, ctx.mul_relin(a, b)), [0, 3, 2, 4, 50, 189])
    check_padded(dec(ctx, a), D1)  # operands unchanged
    check_padded(dec(ctx, b), D2)


def test_plain_ops(ctx, oracle_mod):  # :247-336
    a = enc(ctx, oracle_mod, D1, 1)
    pl = ctx.encode(oracle_mod.expand_vector(D2, ctx.n))
    check_padded(dec(ctx, ctx.add_plain(a, pl)), [3, 4, 3, 5, 15, 30])
    check_padded(dec(ctx, ctx.sub_plain(a, pl)), [3, 2, -1, 3, -5, -12])
    check_padded(dec(ctx, ctx.multiply_plain(a, pl)), [0, 3, 2, 4, 50, 189])


@pytest.mark.parametrize("steps", [4])
def test_rotate_lhs(ctx, oracle_mod, steps):  # rotateCiphertextLhs / Inplace :51-88,:117-140
    ct = enc(ctx, oracle_mod, ROT, 3)
    dv = dec(ctx, ctx.rotate(ct, steps))
    row = ctx.n // 2
    n0 = len(ROT)
    for i in range(len(dv)):
        if i < min(n0 - steps, row - steps):
            assert dv[i] == ROT[i + steps]
        elif row - steps <= i < row:
            assert dv[i] == ROT[i - (row - steps)]
        else:
            assert dv[i] == ROT[-1]
    check_padded(dec(ctx, ct), ROT)


def test_rotate_rhs(ctx, oracle_mod):  # rotateCiphertextRhs :90-115
    steps = -24
    ct = enc(ctx, oracle_mod, ROT, 3)
    dv = dec(ctx, ctx.rotate(ct, steps))
    for i in range(len(dv)):
        if i < abs(steps) or i >= abs(steps) + len(ROT):
            assert dv[i] == ROT[-1]
        else:
            assert dv[i] == ROT[i + steps]


# ---- RuntimeVisitorTest.cpp programs, expressed as the op sequence SpecialRuntimeVisitor issues ----
IN0 = [43, 1, 1, 1, 22, 11, 425, 0, 1, 7]


def test_rv_rotate_negative(ctx, oracle_mod):  # testRotateNegative :67-107
    y = dec(ctx, ctx.rotate(enc(ctx, oracle_mod, IN0, 5), -4))
    assert list(y[:14]) == [7, 7, 7, 7, 43, 1, 1, 1, 22, 11, 425, 0, 1, 7]


def test_rv_rotate_positive(ctx, oracle_mod):  # :509-547
    y = dec(ctx, ctx.rotate(enc(ctx, oracle_mod, IN0, 5), 6))
    assert list(y[:9]) == [425, 0, 1, 7, 7, 7, 7, 7, 7]


def test_rv_ctxt_ctxt(ctx, oracle_mod):  # testBinaryExpressionCtxtCtxt :224-262
    a = enc(ctx, oracle_mod, IN0, 5)
    b = enc(ctx, oracle_mod, [24, 34, 222, 4, 1, 4, 9, 22, 1, 3], 6)
    assert list(dec(ctx, ctx.mul_relin(a, b))[:10]) == [1032, 34, 222, 4, 22, 44, 3825, 0, 1, 21]


def test_rv_ctxt_plain_both_orders(ctx, oracle_mod):  # :264-342 ; x = result[3] -> rotateRows(3), slot 0
    a = enc(ctx, oracle_mod, [43, 1, 1, 22, 11, 7], 7)
    pl = ctx.encode(oracle_mod.expand_vector([19], ctx.n))
    r = ctx.multiply_plain(a, pl)
    assert list(dec(ctx, r)[:6]) == [817, 19, 19, 418, 209, 133]
    assert dec(ctx, ctx.rotate(r, 3))[0] == 418


def test_rv_for_loop_ten_adds(ctx, oracle_mod):  # testForLoop :549-594
    a = enc(ctx, oracle_mod, IN0, 5)
    acc = a
    for _ in range(9):
        acc = ctx.add(acc, a)
    assert list(dec(ctx, acc)[:10]) == [430, 10, 10, 10, 220, 110, 4250, 0, 10, 70]


def test_negate_is_multiply_by_minus_one(ctx, oracle_mod):  # SealCiphertext.cpp:156-157,192-193
    a = enc(ctx, oracle_mod, D1, 1)
    check_padded(dec(ctx, ctx.negate(a)), [-v for v in D1])
    pl = ctx.encode(oracle_mod.expand_vector([-1], ctx.n))
    check_padded(dec(ctx, ctx.multiply_plain(a, pl)), [-v for v in D1])


def test_expand_vector_too_long(oracle_mod):  # SealCiphertextFactory.cpp:106-110
    with pytest.raises(RuntimeError):
        oracle_mod.expand_vector([1] * 4097, 4096)


def test_noise_budget_positive_after_multiply(ctx, oracle_mod):  # SealCiphertext.cpp:80-83
    a, b = enc(ctx, oracle_mod, D1, 1), enc(ctx, oracle_mod, D2, 2)
    fresh, after = ctx.noise_budget(a), ctx.noise_budget(ctx.mul_relin(a, b))
    assert fresh > after > 0
