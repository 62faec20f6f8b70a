"""HIP-graph capture of an operation sequence (abc_hip_graph_*): a recorded circuit replays with results
bit-identical to eager execution and to the oracle."""
import ctypes as C
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_captured_circuit_matches_eager_and_oracle(oracle_mod, capi):
    n = 4096
    o = oracle_mod.Oracle.bfv_default(n)
    o.keygen(77)
    g = capi.Context.bfv_default(n)
    g.keygen(77)
    a = o.encrypt(o.encode(oracle_mod.expand_vector([3, 3, 1, 4, 5, 9], n)), 1)
    b = o.encrypt(o.encode(oracle_mod.expand_vector([0, 1, 2, 1, 10, 21], n)), 2)
    da, db = g.upload(a), g.upload(b)
    t1, t2, out = g.alloc(a.nbytes), g.alloc(a.nbytes), g.alloc(a.nbytes)
    L, one = g.L, C.c_size_t(1)

    def circuit():  # r = (a *** b); r = rotate(r, 1) +++ a
        g.op("mul_relin", da.ptr, db.ptr, t1.ptr, L, one)
        g.op("rotate", t1.ptr, t2.ptr, L, 1, one)
        g.op("add", t2.ptr, da.ptr, out.ptr, 2, L, one)

    circuit()  # eager warm-up sizes every scratch arena
    g.sync()
    eager = g.download(out, a.shape)
    want = o.add(o.rotate(o.mul_relin(a, b), 1), a)
    assert np.array_equal(eager, want)

    g.graph_begin()
    circuit()
    graph = g.graph_end()
    # overwrite the output, replay, compare
    g.op("negate", out.ptr, out.ptr, 2, L, one)
    g.graph_launch(graph)
    g.sync()
    assert np.array_equal(g.download(out, a.shape), want)

    reps = 50
    g.sync(); t0 = time.perf_counter()
    for _ in range(reps):
        circuit()
    g.sync(); t_eager = (time.perf_counter() - t0) / reps
    g.sync(); t0 = time.perf_counter()
    for _ in range(reps):
        g.graph_launch(graph)
    g.sync(); t_graph = (time.perf_counter() - t0) / reps
    print("BFV N=4096 mul_relin+rotate+add: eager %.1f us, graph replay %.1f us" % (t_eager * 1e6, t_graph * 1e6))
    assert np.array_equal(g.download(out, a.shape), want)
    g.graph_destroy(graph)


def test_capture_refuses_to_grow_scratch(capi):
    n = 4096
    g = capi.Context.bfv_default(n)
    g.keygen(5)
    x = np.zeros((2, g.L, n), dtype=np.uint64)
    d = g.upload(x)
    out = g.alloc(x.nbytes)
    g.graph_begin()
    with pytest.raises(capi.AbcHipError):  # nothing ran eagerly yet: the key-switch scratch does not exist
        g.op("rotate", d.ptr, out.ptr, g.L, 1, C.c_size_t(1))
    try:
        g.graph_end()
    except capi.AbcHipError:
        pass


def test_captured_circuit_on_the_default_ring_with_lanes(oracle_mod, capi):
    """BFVDefault(16384), a batch large enough for two internal lanes: the split BEHZ multiply (abc_kernels_bmul.hip) and the key
    switch record into a graph with their lane fork / join events; the replay must equal eager execution and the oracle."""
    n = 16384
    o = oracle_mod.Oracle.bfv_default(n)
    o.keygen(78)
    g = capi.Context.bfv_default(n)
    g.keygen(78)
    a = o.encrypt(o.encode(oracle_mod.expand_vector([3, 3, 1, 4, 5, 9], n)), 1)
    b = o.encrypt(o.encode(oracle_mod.expand_vector([0, 1, 2, 1, 10, 21], n)), 2)
    B = 12
    batch_a = np.stack([a if i % 2 else b for i in range(B)])
    batch_b = np.stack([b if i % 3 else a for i in range(B)])
    da, db = g.upload(batch_a), g.upload(batch_b)
    t1, out = g.alloc(batch_a.nbytes), g.alloc(batch_a.nbytes)
    L, cb = g.L, C.c_size_t(B)

    def circuit():
        g.op("mul_relin", da.ptr, db.ptr, t1.ptr, L, cb)
        g.op("rotate", t1.ptr, out.ptr, L, 3, cb)

    circuit()
    g.sync()
    eager = g.download(out, batch_a.shape)
    for i in (0, 1, 5, B - 1):
        assert np.array_equal(eager[i], o.rotate(o.mul_relin(batch_a[i], batch_b[i]), 3)), i
    g.graph_begin()
    circuit()
    graph = g.graph_end()
    g.op("negate", out.ptr, out.ptr, 2, L, cb)
    g.graph_launch(graph)
    g.sync()
    assert np.array_equal(g.download(out, batch_a.shape), eager)
    g.graph_destroy(graph)
