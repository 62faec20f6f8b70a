"""CPU-side boundary checks: libabc_hip.so loads and exports every symbol include/abc_hip.h declares,
and the no-GPU / bad-argument error paths fail loudly (no compute is attempted without a GPU)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "abc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(abc_hip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(capi):
    lib = capi.lib()
    names = header_symbols()
    assert len(names) >= 45
    for n in names:
        assert hasattr(lib, n), "missing export: " + n
    assert sorted(capi.SYMBOLS) == names


def test_host_only_helpers_work_without_gpu(capi):
    assert capi.default_bfv_primes(4096) == [0xffffee001, 0xffffc4001, 0x1ffffe0001]
    assert capi.plain_modulus_batching(4096, 20) == 1032193
    assert capi.create_primes(16384, [50, 40, 40, 40, 50])[0] < capi.create_primes(16384, [50, 40, 40, 40, 50])[4]
    with pytest.raises(capi.AbcHipError):
        capi.default_bfv_primes(1000)


def test_product_and_oracle_agree_on_parameters(capi, oracle_mod):
    for n in (4096, 8192, 16384, 32768):
        assert capi.default_bfv_primes(n) == oracle_mod.default_bfv_primes(n)
        assert capi.plain_modulus_batching(n, 20) == oracle_mod.plain_modulus_batching(n, 20)
    assert capi.create_primes(32768, [55, 55, 40, 56]) == oracle_mod.create_primes(32768, [55, 55, 40, 56])


def test_context_creation_fails_loudly(capi):
    lib = capi.lib()
    if lib.abc_hip_device_count() > 0:
        pytest.skip("GPU present: the no-device error path is not reachable")
    with pytest.raises(capi.AbcHipError) as e:
        capi.Context.bfv_default(4096)
    assert "no HIP device" in str(e.value)


def test_bad_parameters_rejected(capi):
    with pytest.raises(capi.AbcHipError):
        capi.Context(capi.BFV, 4096, [97, 193, 257], 1032193)       # not = 1 mod 2N
    with pytest.raises(capi.AbcHipError):
        capi.Context(capi.BFV, 4096, capi.default_bfv_primes(4096), 1032191)  # t not a batching prime
    with pytest.raises(capi.AbcHipError):
        capi.Context(7, 4096, capi.default_bfv_primes(4096), 1032193)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "abc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_py" not in text and "liboracle" not in text and "orc_" not in text, f
