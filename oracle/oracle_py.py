"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

The oracle is the CPU restatement of the SEAL 3.6 algorithms behind the reference's
SealCiphertext / SealCiphertextFactory (src/runtime/SealCiphertext.cpp,
src/runtime/SealCiphertextFactory.cpp).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the product (abc_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

BFV, CKKS = 1, 2


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    stale = (not os.path.exists(_LIB)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None
u64p = C.POINTER(C.c_uint64)
i64p = C.POINTER(C.c_int64)
dblp = C.POINTER(C.c_double)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.orc_ctx_create.restype = C.c_void_p
        L.orc_ctx_create.argtypes = [C.c_int, C.c_int, u64p, C.c_int, C.c_uint64]
        L.orc_ctx_destroy.argtypes = [C.c_void_p]
        L.orc_ctx_prime.restype = C.c_uint64
        L.orc_ctx_prime.argtypes = [C.c_void_p, C.c_int]
        L.orc_ctx_info.argtypes = [C.c_void_p, C.c_int]
        L.orc_ctx_plain_modulus.restype = C.c_uint64
        L.orc_ctx_plain_modulus.argtypes = [C.c_void_p]
        L.orc_ctx_ntt_root.restype = C.c_uint64
        L.orc_ctx_ntt_root.argtypes = [C.c_void_p, C.c_int]
        L.orc_ctx_behz_prime.restype = C.c_uint64
        L.orc_ctx_behz_prime.argtypes = [C.c_void_p, C.c_int]
        L.orc_plain_modulus_batching.restype = C.c_uint64
        L.orc_plain_modulus_batching.argtypes = [C.c_size_t, C.c_int]
        L.orc_default_bfv_primes.argtypes = [C.c_size_t, u64p]
        L.orc_create_primes.argtypes = [C.c_size_t, C.POINTER(C.c_int), C.c_int, u64p]
        L.orc_is_prime.argtypes = [C.c_uint64]
        L.orc_galois_elt_from_step.restype = C.c_uint32
        L.orc_galois_elt_from_step.argtypes = [C.c_void_p, C.c_int]
        L.orc_galois_elt_at.restype = C.c_uint32
        L.orc_galois_elt_at.argtypes = [C.c_void_p, C.c_int]
        L.orc_time_mul_relin.restype = C.c_double
        _lib = L
    return _lib


def _p(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


def default_bfv_primes(n):
    out = np.zeros(16, dtype=np.uint64)
    cnt = lib().orc_default_bfv_primes(n, _p(out))
    if cnt < 0:
        raise ValueError("no default BFV modulus for N=%d" % n)
    return [int(x) for x in out[:cnt]]


def create_primes(n, bit_sizes):
    out = np.zeros(len(bit_sizes), dtype=np.uint64)
    bs = (C.c_int * len(bit_sizes))(*bit_sizes)
    if lib().orc_create_primes(n, bs, len(bit_sizes), _p(out)):
        raise ValueError("cannot create primes")
    return [int(x) for x in out]


def plain_modulus_batching(n, bits=20):
    return int(lib().orc_plain_modulus_batching(n, bits))


class Oracle:
    """One encryption context (params + keys) of the CPU oracle."""

    def __init__(self, scheme, n, primes, plain_modulus=0):
        self.scheme, self.n = scheme, n
        self.logn = n.bit_length() - 1
        self.primes = list(primes)
        arr = np.array(primes, dtype=np.uint64)
        self.h = lib().orc_ctx_create(scheme, self.logn, _p(arr), len(primes), plain_modulus)
        if not self.h:
            raise ValueError("orc_ctx_create failed")
        self.h = C.c_void_p(self.h)
        self.K = len(primes)
        self.L = self.K - 1
        self.t = plain_modulus

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_ctx_destroy(self.h)
            self.h = None

    @classmethod
    def bfv_default(cls, n):
        """Parameters of SealCiphertextFactory::setupSealContext (SealCiphertextFactory.cpp:72-100)."""
        return cls(BFV, n, default_bfv_primes(n), plain_modulus_batching(n, 20))

    # ---- keys ----
    def keygen(self, seed):
        assert lib().orc_keygen(self.h, C.c_uint64(seed)) == 0

    def secret_key(self):
        out = np.zeros((self.K, self.n), dtype=np.uint64)
        assert lib().orc_get_secret_key(self.h, _p(out)) == 0
        return out

    def public_key(self):
        out = np.zeros((2, self.K, self.n), dtype=np.uint64)
        assert lib().orc_get_public_key(self.h, _p(out)) == 0
        return out

    def relin_key(self):
        out = np.zeros((self.L, 2, self.K, self.n), dtype=np.uint64)
        assert lib().orc_get_relin_key(self.h, _p(out)) == 0
        return out

    def galois_elts(self):
        return [int(lib().orc_galois_elt_at(self.h, i)) for i in range(lib().orc_num_galois(self.h))]

    def galois_key(self, elt):
        out = np.zeros((self.L, 2, self.K, self.n), dtype=np.uint64)
        assert lib().orc_get_galois_key(self.h, C.c_uint32(elt), _p(out)) == 0
        return out

    def elt_from_step(self, step):
        return int(lib().orc_galois_elt_from_step(self.h, step))

    # ---- raw transforms ----
    def ntt(self, prime_index, a):
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        lib().orc_ntt_forward(self.h, prime_index, _p(a))
        return a

    def intt(self, prime_index, a):
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        lib().orc_ntt_inverse(self.h, prime_index, _p(a))
        return a

    # ---- BFV ----
    def encode(self, values):
        v = np.ascontiguousarray(values, dtype=np.int64)
        out = np.zeros(self.n, dtype=np.uint64)
        assert lib().orc_batch_encode(self.h, v.ctypes.data_as(i64p), C.c_size_t(len(v)), _p(out)) == 0
        return out

    def decode(self, plain):
        out = np.zeros(self.n, dtype=np.int64)
        assert lib().orc_batch_decode(self.h, _p(plain), out.ctypes.data_as(i64p)) == 0
        return out

    def encrypt(self, plain, seed):
        ct = np.zeros((2, self.L, self.n), dtype=np.uint64)
        fn = lib().orc_bfv_encrypt if self.scheme == BFV else lib().orc_ckks_encrypt
        assert fn(self.h, _p(plain), C.c_uint64(seed), _p(ct)) == 0
        return ct

    def decrypt(self, ct):
        ct = np.ascontiguousarray(ct)
        if self.scheme == BFV:
            out = np.zeros(self.n, dtype=np.uint64)
            assert lib().orc_bfv_decrypt(self.h, _p(ct), ct.shape[0], _p(out)) == 0
        else:
            out = np.zeros((ct.shape[1], self.n), dtype=np.uint64)
            assert lib().orc_ckks_decrypt(self.h, _p(ct), ct.shape[0], ct.shape[1], _p(out)) == 0
        return out

    def noise_budget(self, ct):
        ct = np.ascontiguousarray(ct)
        return lib().orc_bfv_noise_budget(self.h, _p(ct), ct.shape[0])

    def _binop(self, fn, a, b):
        a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
        out = np.zeros_like(a)
        assert fn(self.h, _p(a), _p(b), a.shape[0], a.shape[1], _p(out)) == 0
        return out

    def add(self, a, b):
        return self._binop(lib().orc_add, a, b)

    def sub(self, a, b):
        return self._binop(lib().orc_sub, a, b)

    def negate(self, a):
        a = np.ascontiguousarray(a)
        out = np.zeros_like(a)
        assert lib().orc_negate(self.h, _p(a), a.shape[0], a.shape[1], _p(out)) == 0
        return out

    def multiply(self, a, b):
        a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
        out = np.zeros((3,) + a.shape[1:], dtype=np.uint64)
        if self.scheme == BFV:
            assert lib().orc_bfv_multiply(self.h, _p(a), _p(b), _p(out)) == 0
        else:
            assert lib().orc_ckks_multiply(self.h, _p(a), _p(b), a.shape[1], _p(out)) == 0
        return out

    def relinearize(self, ct3):
        ct3 = np.ascontiguousarray(ct3)
        out = np.zeros((2,) + ct3.shape[1:], dtype=np.uint64)
        assert lib().orc_relinearize(self.h, _p(ct3), ct3.shape[1], _p(out)) == 0
        return out

    def mul_relin(self, a, b):
        return self.relinearize(self.multiply(a, b))

    def rotate(self, ct, steps):
        ct = np.ascontiguousarray(ct)
        out = np.zeros_like(ct)
        rc = lib().orc_rotate(self.h, _p(ct), ct.shape[1], steps, _p(out))
        if rc:
            raise RuntimeError("orc_rotate rc=%d" % rc)
        return out

    def apply_galois(self, ct, elt):
        ct = np.ascontiguousarray(ct)
        out = np.zeros_like(ct)
        assert lib().orc_apply_galois(self.h, _p(ct), ct.shape[1], C.c_uint32(elt), _p(out)) == 0
        return out

    def galois_permute(self, poly, elt, ntt_form):
        poly = np.ascontiguousarray(poly)
        out = np.zeros_like(poly)
        assert lib().orc_galois_permute(self.h, _p(poly), poly.shape[0], C.c_uint32(elt), int(ntt_form), _p(out)) == 0
        return out

    def keyswitch(self, target, key):
        target, key = np.ascontiguousarray(target), np.ascontiguousarray(key)
        out = np.zeros((2,) + target.shape, dtype=np.uint64)
        assert lib().orc_keyswitch(self.h, _p(target), target.shape[0], _p(key), _p(out)) == 0
        return out

    def multiply_plain(self, ct, plain):
        ct = np.ascontiguousarray(ct)
        out = np.zeros_like(ct)
        if self.scheme == BFV:
            assert lib().orc_bfv_multiply_plain(self.h, _p(ct), ct.shape[0], _p(plain), _p(out)) == 0
        else:
            assert lib().orc_ckks_multiply_plain(self.h, _p(ct), ct.shape[0], ct.shape[1], _p(plain), _p(out)) == 0
        return out

    def add_plain(self, ct, plain):
        ct = np.ascontiguousarray(ct)
        out = np.zeros_like(ct)
        if self.scheme == BFV:
            assert lib().orc_bfv_add_plain(self.h, _p(ct), ct.shape[0], _p(plain), _p(out)) == 0
        else:
            assert lib().orc_ckks_add_plain(self.h, _p(ct), ct.shape[0], ct.shape[1], _p(plain), _p(out)) == 0
        return out

    def sub_plain(self, ct, plain):
        ct = np.ascontiguousarray(ct)
        out = np.zeros_like(ct)
        assert lib().orc_bfv_sub_plain(self.h, _p(ct), ct.shape[0], _p(plain), _p(out)) == 0
        return out

    # ---- CKKS ----
    def ckks_encode(self, values, scale, nl=None):
        nl = self.L if nl is None else nl
        v = np.asarray(values, dtype=np.complex128)
        re = np.ascontiguousarray(v.real)
        im = np.ascontiguousarray(v.imag)
        out = np.zeros((nl, self.n), dtype=np.uint64)
        rc = lib().orc_ckks_encode(self.h, re.ctypes.data_as(dblp), im.ctypes.data_as(dblp), C.c_size_t(len(v)),
                                   C.c_double(scale), nl, _p(out))
        assert rc == 0, rc
        return out

    def ckks_decode(self, plain, scale):
        plain = np.ascontiguousarray(plain)
        re = np.zeros(self.n // 2)
        im = np.zeros(self.n // 2)
        assert lib().orc_ckks_decode(self.h, _p(plain), plain.shape[0], C.c_double(scale), re.ctypes.data_as(dblp),
                                     im.ctypes.data_as(dblp)) == 0
        return re + 1j * im

    def rescale(self, ct):
        ct = np.ascontiguousarray(ct)
        out = np.zeros((ct.shape[0], ct.shape[1] - 1, self.n), dtype=np.uint64)
        assert lib().orc_ckks_rescale(self.h, _p(ct), ct.shape[0], ct.shape[1], _p(out)) == 0
        return out

    def mod_switch(self, ct):
        ct = np.ascontiguousarray(ct)
        out = np.zeros((ct.shape[0], ct.shape[1] - 1, self.n), dtype=np.uint64)
        assert lib().orc_ckks_mod_switch(self.h, _p(ct), ct.shape[0], ct.shape[1], _p(out)) == 0
        return out

    def time_mul_relin(self, a, b, iters):
        a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
        out = np.zeros_like(a)
        return float(lib().orc_time_mul_relin(self.h, _p(a), _p(b), a.shape[1], iters, _p(out)))


def expand_vector(values, n):
    """SealCiphertextFactory::expandVector (SealCiphertextFactory.cpp:102-115): pad with the last value."""
    v = list(values)
    if len(v) > n:
        raise RuntimeError("Cannot encode %d elements in a ciphertext of size %d. " % (len(v), n))
    return v + [v[-1]] * (n - len(v))
