"""ctypes binding of libabc_hip.so (the C ABI declared in include/abc_hip.h).

Plumbing for tests, bench.py and __graft_entry__: every call goes through the C ABI into the HIP
kernels.  There is NO CPU fallback: a missing library, a missing GPU or a failing call raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ABC_HIP_LIB") or os.path.join(_HERE, "libabc_hip.so")  # override: A/B of two builds

BFV, CKKS = 1, 2
u64p = C.POINTER(C.c_uint64)
i64p = C.POINTER(C.c_int64)

_lib = None

# every symbol include/abc_hip.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "abc_hip_last_error", "abc_hip_device_count", "abc_hip_ctx_create", "abc_hip_ctx_destroy",
    "abc_hip_default_bfv_primes", "abc_hip_plain_modulus_batching", "abc_hip_create_primes", "abc_hip_ctx_info",
    "abc_hip_set_stream", "abc_hip_sync", "abc_hip_ctx_reload_env", "abc_hip_malloc", "abc_hip_free", "abc_hip_trim",
    "abc_hip_cached_bytes", "abc_hip_memcpy_h2d", "abc_hip_memcpy_d2h",
    "abc_hip_memcpy_d2d", "abc_hip_keygen", "abc_hip_keygen_secure", "abc_hip_encrypt_secure", "abc_hip_load_secret_key", "abc_hip_load_public_key",
    "abc_hip_load_relin_key", "abc_hip_load_galois_key", "abc_hip_get_secret_key", "abc_hip_get_public_key",
    "abc_hip_get_relin_key", "abc_hip_get_galois_key", "abc_hip_num_galois_keys", "abc_hip_galois_elt_at",
    "abc_hip_galois_elt_from_step", "abc_hip_batch_encode", "abc_hip_batch_decode", "abc_hip_encrypt", "abc_hip_decrypt",
    "abc_hip_add", "abc_hip_sub", "abc_hip_negate", "abc_hip_multiply", "abc_hip_relinearize", "abc_hip_mul_relin",
    "abc_hip_rotate", "abc_hip_apply_galois", "abc_hip_multiply_plain", "abc_hip_add_plain", "abc_hip_sub_plain",
    "abc_hip_rescale", "abc_hip_mod_switch", "abc_hip_ntt_forward", "abc_hip_ntt_inverse", "abc_hip_keyswitch",
    "abc_hip_microbench", "abc_hip_timer_start", "abc_hip_timer_stop",
    "abc_hip_ntt_limbs",
    "abc_hip_graph_begin", "abc_hip_graph_end", "abc_hip_graph_launch", "abc_hip_graph_destroy",
]


class AbcHipError(RuntimeError):
    pass


def lib():
    """Load libabc_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AbcHipError("libabc_hip.so is not built: run `python -m abc_amd.build` (hipcc, gfx950)")
        L = C.CDLL(LIB_PATH)
        L.abc_hip_last_error.restype = C.c_char_p
        L.abc_hip_plain_modulus_batching.restype = C.c_uint64
        L.abc_hip_plain_modulus_batching.argtypes = [C.c_size_t, C.c_int]
        L.abc_hip_default_bfv_primes.argtypes = [C.c_size_t, u64p]
        L.abc_hip_create_primes.argtypes = [C.c_size_t, C.POINTER(C.c_int), C.c_int, u64p]
        L.abc_hip_ctx_create.argtypes = [C.c_int, C.c_int, u64p, C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_void_p)]
        L.abc_hip_ctx_destroy.argtypes = [C.c_void_p]
        L.abc_hip_galois_elt_at.restype = C.c_uint32
        L.abc_hip_cached_bytes.restype = C.c_size_t
        L.abc_hip_cached_bytes.argtypes = [C.c_void_p]
        L.abc_hip_galois_elt_from_step.restype = C.c_uint32
        _lib = L
    return _lib


def _chk(rc):
    if rc != 0:
        raise AbcHipError(lib().abc_hip_last_error().decode())


def default_bfv_primes(n):
    out = (C.c_uint64 * 16)()
    cnt = lib().abc_hip_default_bfv_primes(n, out)
    if cnt < 0:
        raise AbcHipError(lib().abc_hip_last_error().decode())
    return [int(out[i]) for i in range(cnt)]


def plain_modulus_batching(n, bits=20):
    v = int(lib().abc_hip_plain_modulus_batching(n, bits))
    if not v:
        raise AbcHipError(lib().abc_hip_last_error().decode())
    return v


def create_primes(n, bit_sizes):
    out = (C.c_uint64 * len(bit_sizes))()
    bs = (C.c_int * len(bit_sizes))(*bit_sizes)
    _chk(lib().abc_hip_create_primes(n, bs, len(bit_sizes), out))
    return [int(x) for x in out]


class DeviceBuffer:
    """A device allocation owned through the C ABI."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = C.c_void_p()
        _chk(lib().abc_hip_malloc(ctx.h, C.byref(p), C.c_size_t(self.nbytes)))
        self.ptr = p

    def free(self):
        if self.ptr is not None and self.ctx.h:
            lib().abc_hip_free(self.ctx.h, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One abc_hip_ctx: parameters, device tables and keys on one GPU."""

    def __init__(self, scheme, n, primes, plain_modulus=0, device=0):
        self.scheme, self.n = scheme, n
        self.logn = n.bit_length() - 1
        self.primes = list(primes)
        self.K, self.L = len(primes), len(primes) - 1
        self.t = plain_modulus
        arr = (C.c_uint64 * len(primes))(*primes)
        h = C.c_void_p()
        _chk(lib().abc_hip_ctx_create(scheme, self.logn, arr, len(primes), C.c_uint64(plain_modulus), device, C.byref(h)))
        self.h = h

    @classmethod
    def bfv_default(cls, n, device=0):
        """The parameters SealCiphertextFactory::setupSealContext picks (SealCiphertextFactory.cpp:72-100)."""
        return cls(BFV, n, default_bfv_primes(n), plain_modulus_batching(n, 20), device)

    def close(self):
        if getattr(self, "h", None):
            lib().abc_hip_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- memory ----
    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        buf = DeviceBuffer(self, arr.nbytes)
        _chk(lib().abc_hip_memcpy_h2d(self.h, buf.ptr, arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes)))
        return buf

    def download(self, buf, shape, dtype=np.uint64):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= buf.nbytes
        _chk(lib().abc_hip_memcpy_d2h(self.h, out.ctypes.data_as(C.c_void_p), buf.ptr, C.c_size_t(out.nbytes)))
        return out

    def sync(self):
        _chk(lib().abc_hip_sync(self.h))

    def trim(self):
        _chk(lib().abc_hip_trim(self.h))

    def cached_bytes(self):
        return int(lib().abc_hip_cached_bytes(self.h))

    def reload_env(self):
        _chk(lib().abc_hip_ctx_reload_env(self.h))

    def set_stream(self, stream_ptr):
        _chk(lib().abc_hip_set_stream(self.h, C.c_void_p(stream_ptr)))

    # ---- keys ----
    def keygen(self, seed=None):
        """seed=None: OS-keyed ChaCha20 (deployment); an integer: the reproducible test sampling spec shared with the oracle"""
        if seed is None:
            _chk(lib().abc_hip_keygen_secure(self.h))
        else:
            _chk(lib().abc_hip_keygen(self.h, C.c_uint64(seed)))

    def load_keys(self, sk=None, pk=None, relin=None, galois=None):
        def p(a):
            a = np.ascontiguousarray(a, dtype=np.uint64)
            return a, a.ctypes.data_as(u64p)
        if sk is not None:
            a, ptr = p(sk); _chk(lib().abc_hip_load_secret_key(self.h, ptr))
        if pk is not None:
            a, ptr = p(pk); _chk(lib().abc_hip_load_public_key(self.h, ptr))
        if relin is not None:
            a, ptr = p(relin); _chk(lib().abc_hip_load_relin_key(self.h, ptr))
        for elt, key in (galois or {}).items():
            a, ptr = p(key); _chk(lib().abc_hip_load_galois_key(self.h, C.c_uint32(elt), ptr))

    def get_key(self, which, elt=0):
        shape = {"sk": (self.K, self.n), "pk": (2, self.K, self.n)}.get(which, (self.L, 2, self.K, self.n))
        out = np.zeros(shape, dtype=np.uint64)
        ptr = out.ctypes.data_as(u64p)
        if which == "sk":
            _chk(lib().abc_hip_get_secret_key(self.h, ptr))
        elif which == "pk":
            _chk(lib().abc_hip_get_public_key(self.h, ptr))
        elif which == "relin":
            _chk(lib().abc_hip_get_relin_key(self.h, ptr))
        else:
            _chk(lib().abc_hip_get_galois_key(self.h, C.c_uint32(elt), ptr))
        return out

    def galois_elts(self):
        return [int(lib().abc_hip_galois_elt_at(self.h, i)) for i in range(lib().abc_hip_num_galois_keys(self.h))]

    def elt_from_step(self, step):
        return int(lib().abc_hip_galois_elt_from_step(self.h, step))

    # ---- raw device-pointer ops (count = batch) ----
    def op(self, name, *args):
        _chk(getattr(lib(), "abc_hip_" + name)(self.h, *args))

    # ---- numpy convenience: host array in -> device op -> host array out ----
    def _run(self, name, ins, out_shape, *scalars, out_dtype=np.uint64):
        bufs = [self.upload(a) for a in ins]
        out = self.alloc(int(np.prod(out_shape)) * np.dtype(out_dtype).itemsize)
        self.op(name, *[b.ptr for b in bufs], out.ptr, *scalars)
        res = self.download(out, out_shape, out_dtype)
        for b in bufs + [out]:
            b.free()
        return res

    @staticmethod
    def _batch(a, ndim):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        return a if a.ndim == ndim + 1 else a[None]

    def add(self, a, b):
        a4, b4 = self._batch(a, 3), self._batch(b, 3)
        r = self._run("add", [a4, b4], a4.shape, a4.shape[1], a4.shape[2], C.c_size_t(a4.shape[0]))
        return r if np.ndim(a) == 4 else r[0]

    def sub(self, a, b):
        a4, b4 = self._batch(a, 3), self._batch(b, 3)
        r = self._run("sub", [a4, b4], a4.shape, a4.shape[1], a4.shape[2], C.c_size_t(a4.shape[0]))
        return r if np.ndim(a) == 4 else r[0]

    def negate(self, a):
        a4 = self._batch(a, 3)
        r = self._run("negate", [a4], a4.shape, a4.shape[1], a4.shape[2], C.c_size_t(a4.shape[0]))
        return r if np.ndim(a) == 4 else r[0]

    def multiply(self, a, b):
        a4, b4 = self._batch(a, 3), self._batch(b, 3)
        shp = (a4.shape[0], 3) + a4.shape[2:]
        r = self._run("multiply", [a4, b4], shp, a4.shape[2], C.c_size_t(a4.shape[0]))
        return r if np.ndim(a) == 4 else r[0]

    def relinearize(self, ct3):
        a4 = self._batch(ct3, 3)
        shp = (a4.shape[0], 2) + a4.shape[2:]
        r = self._run("relinearize", [a4], shp, a4.shape[2], C.c_size_t(a4.shape[0]))
        return r if np.ndim(ct3) == 4 else r[0]

    def mul_relin(self, a, b):
        a4, b4 = self._batch(a, 3), self._batch(b, 3)
        r = self._run("mul_relin", [a4, b4], a4.shape, a4.shape[2], C.c_size_t(a4.shape[0]))
        return r if np.ndim(a) == 4 else r[0]

    def rotate(self, ct, steps):
        a4 = self._batch(ct, 3)
        r = self._run("rotate", [a4], a4.shape, a4.shape[2], int(steps), C.c_size_t(a4.shape[0]))
        return r if np.ndim(ct) == 4 else r[0]

    def apply_galois(self, ct, elt):
        a4 = self._batch(ct, 3)
        r = self._run("apply_galois", [a4], a4.shape, a4.shape[2], C.c_uint32(elt), C.c_size_t(a4.shape[0]))
        return r if np.ndim(ct) == 4 else r[0]

    def _plain_op(self, name, ct, plain):
        a4 = self._batch(ct, 3)
        plain = np.ascontiguousarray(plain, dtype=np.uint64)
        per = self.n if self.scheme == BFV else a4.shape[2] * self.n
        stride = per if plain.size == a4.shape[0] * per and a4.shape[0] > 1 else 0
        if a4.shape[0] == 1:
            stride = 0
        ctb, plb = self.upload(a4), self.upload(plain)
        out = self.alloc(a4.nbytes)
        self.op(name, ctb.ptr, plb.ptr, C.c_size_t(stride), out.ptr, a4.shape[1], a4.shape[2], C.c_size_t(a4.shape[0]))
        r = self.download(out, a4.shape)
        for b in (ctb, plb, out):
            b.free()
        return r if np.ndim(ct) == 4 else r[0]

    def multiply_plain(self, ct, plain):
        return self._plain_op("multiply_plain", ct, plain)

    def add_plain(self, ct, plain):
        return self._plain_op("add_plain", ct, plain)

    def sub_plain(self, ct, plain):
        return self._plain_op("sub_plain", ct, plain)

    def rescale(self, ct):
        a4 = self._batch(ct, 3)
        shp = (a4.shape[0], a4.shape[1], a4.shape[2] - 1, self.n)
        r = self._run("rescale", [a4], shp, a4.shape[1], a4.shape[2], C.c_size_t(a4.shape[0]))
        return r if np.ndim(ct) == 4 else r[0]

    def mod_switch(self, ct):
        a4 = self._batch(ct, 3)
        shp = (a4.shape[0], a4.shape[1], a4.shape[2] - 1, self.n)
        r = self._run("mod_switch", [a4], shp, a4.shape[1], a4.shape[2], C.c_size_t(a4.shape[0]))
        return r if np.ndim(ct) == 4 else r[0]

    def keyswitch(self, target, key_kind):
        t3 = self._batch(target, 2)
        shp = (t3.shape[0], 2) + t3.shape[1:]
        tb = self.upload(t3)
        out = self.alloc(int(np.prod(shp)) * 8)
        self.op("keyswitch", tb.ptr, C.c_uint32(key_kind), out.ptr, t3.shape[1], C.c_size_t(t3.shape[0]))
        r = self.download(out, shp)
        tb.free(); out.free()
        return r if np.ndim(target) == 3 else r[0]

    def ntt(self, data, kind, index, inverse=False):
        d = np.ascontiguousarray(data, dtype=np.uint64)
        d2 = d.reshape(-1, self.n)
        buf = self.upload(d2)
        self.op("ntt_inverse" if inverse else "ntt_forward", buf.ptr, kind, index, C.c_size_t(d2.shape[0]))
        r = self.download(buf, d2.shape).reshape(d.shape)
        buf.free()
        return r

    def ntt_limbs(self, data, inverse=False):
        """[polys][nl][N] coefficient form <-> NTT form at a data level (limb j modulo q_j)."""
        d = np.ascontiguousarray(data, dtype=np.uint64)
        d3 = d.reshape(-1, d.shape[-2], self.n)
        buf = self.upload(d3)
        self.op("ntt_limbs", buf.ptr, d3.shape[1], C.c_size_t(d3.shape[0]), 1 if inverse else 0)
        r = self.download(buf, d3.shape).reshape(d.shape)
        buf.free()
        return r

    def batch_encode(self, values):
        v = np.ascontiguousarray(values, dtype=np.int64).reshape(-1, self.n)
        vb = self.upload(v)
        out = self.alloc(v.nbytes)
        self.op("batch_encode", vb.ptr, out.ptr, C.c_size_t(v.shape[0]))
        r = self.download(out, v.shape)
        vb.free(); out.free()
        return r if np.ndim(values) == 2 else r[0]

    def batch_decode(self, plain):
        p = np.ascontiguousarray(plain, dtype=np.uint64).reshape(-1, self.n)
        pb = self.upload(p)
        out = self.alloc(p.nbytes)
        self.op("batch_decode", pb.ptr, out.ptr, C.c_size_t(p.shape[0]))
        r = self.download(out, p.shape, np.int64)
        pb.free(); out.free()
        return r if np.ndim(plain) == 2 else r[0]

    def encrypt(self, plain, seed=None):
        per = (self.n,) if self.scheme == BFV else (self.L, self.n)
        p = np.ascontiguousarray(plain, dtype=np.uint64).reshape((-1,) + per)
        pb = self.upload(p)
        shp = (p.shape[0], 2, self.L, self.n)
        out = self.alloc(int(np.prod(shp)) * 8)
        if seed is None:
            self.op("encrypt_secure", pb.ptr, out.ptr, C.c_size_t(p.shape[0]))
        else:
            self.op("encrypt", pb.ptr, C.c_uint64(seed), out.ptr, C.c_size_t(p.shape[0]))
        r = self.download(out, shp)
        pb.free(); out.free()
        return r if np.ndim(plain) == len(per) + 1 else r[0]

    def decrypt(self, ct):
        a4 = self._batch(ct, 3)
        cb = self.upload(a4)
        shp = (a4.shape[0], self.n) if self.scheme == BFV else (a4.shape[0], a4.shape[2], self.n)
        out = self.alloc(int(np.prod(shp)) * 8)
        self.op("decrypt", cb.ptr, a4.shape[1], a4.shape[2], out.ptr, C.c_size_t(a4.shape[0]))
        r = self.download(out, shp)
        cb.free(); out.free()
        return r if np.ndim(ct) == 4 else r[0]

    # ---- HIP-graph capture of an op sequence ----
    def graph_begin(self):
        _chk(lib().abc_hip_graph_begin(self.h))

    def graph_end(self):
        g = C.c_void_p()
        _chk(lib().abc_hip_graph_end(self.h, C.byref(g)))
        return g

    def graph_launch(self, g):
        _chk(lib().abc_hip_graph_launch(self.h, g))

    def graph_destroy(self, g):
        _chk(lib().abc_hip_graph_destroy(self.h, g))

    def microbench(self, which, iters):
        ms = C.c_double()
        _chk(lib().abc_hip_microbench(self.h, which, iters, C.byref(ms)))
        return ms.value

    def timer_start(self):
        _chk(lib().abc_hip_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float()
        _chk(lib().abc_hip_timer_stop(self.h, C.byref(ms)))
        return ms.value
