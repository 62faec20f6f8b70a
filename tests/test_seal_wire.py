"""SEAL 3.6 wire format (abc_amd/runtime/SealWire.*).

PARITY UNPINNED: the reference serialises nothing and SEAL is not in the image, so no SEAL-produced bytes exist to compare
with.  CPU: the C++ codec's own tests, then an INDEPENDENT reader written here (struct + zlib + hashlib) over the objects
the C++ writer dumps -- it follows SEAL 3.6's published layout separately from SealWire.cpp, so a slip in either shows.
GPU: ciphertexts and secret / public / relinearisation / Galois keys travel between two HipCiphertextFactory instances.
"""
import hashlib
import os
import struct
import subprocess
import zlib

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RT = os.path.join(ROOT, "abc_amd", "runtime")


@pytest.fixture(scope="module")
def cpu_binary():
    subprocess.check_call(["make", "-C", RT, "test_seal_wire_cpu"], stdout=subprocess.DEVNULL)
    return os.path.join(RT, "test_seal_wire_cpu")


def test_codec_blake2b_headers_round_trips_and_refusals(cpu_binary):
    p = subprocess.run([cpu_binary], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert " 0 failed" in p.stdout


# ---- the independent reader ----
def _header(buf, off):
    magic, hsize, major, minor, mode, reserved, size = struct.unpack_from("<HBBBBHQ", buf, off)
    assert magic == 0xA15E and hsize == 16 and (major, minor) == (3, 6) and reserved == 0
    return mode, size


def _object(buf, off):
    """one top-level object -> (body bytes, offset behind it)"""
    mode, size = _header(buf, off)
    raw = buf[off + 16:off + size]
    assert len(raw) == size - 16
    if mode == 1:
        raw = zlib.decompress(raw)
    else:
        assert mode == 0
    return raw, off + size


def _array(body, off):
    mode, size = _header(body, off)
    assert mode == 0
    (count,) = struct.unpack_from("<Q", body, off + 16)
    assert size == 24 + 8 * count
    return list(struct.unpack_from("<%dQ" % count, body, off + 24)), off + size


def _ciphertext(body, off):
    pid = struct.unpack_from("<4Q", body, off)
    ntt, size, n, limbs, scale = struct.unpack_from("<BQQQd", body, off + 32)
    data, off = _array(body, off + 65)
    assert len(data) == size * n * limbs
    return {"id": pid, "ntt": ntt, "size": size, "n": n, "limbs": limbs, "scale": scale, "data": data}, off


def _parms_id(scheme, n, primes, t):
    words = [scheme, n] + list(primes) + ([t] if t else [])
    return struct.unpack("<4Q", hashlib.blake2b(struct.pack("<%dQ" % len(words), *words), digest_size=32).digest())


def test_independent_python_reader_parses_what_the_codec_writes(cpu_binary, tmp_path):
    path = str(tmp_path / "objects.seal")
    p = subprocess.run([cpu_binary, "--dump", path], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    buf = open(path, "rb").read()
    primes = [0xffffee001, 0xffffc4001, 0x1ffffe0001]
    # 1, 2: the same ciphertext, uncompressed and zlib-compressed
    body, off = _object(buf, 0)
    ct, end = _ciphertext(body, 0)
    assert end == len(body)
    assert ct["id"] == _parms_id(1, 64, primes[:2], 65537)  # a ciphertext names ITS level: without the special prime
    assert (ct["size"], ct["n"], ct["limbs"], ct["ntt"]) == (2, 64, 2, 0) and ct["scale"] == 5.0
    body2, off = _object(buf, off)
    assert body2 == body
    # 3: key-switching keys with N entries, entry 1 populated by two public keys, each with its own header
    body, off = _object(buf, off)
    assert struct.unpack_from("<4Q", body, 0) == _parms_id(1, 64, primes, 65537)
    (dim1,) = struct.unpack_from("<Q", body, 32)
    assert dim1 == 64
    pos, filled = 40, {}
    for idx in range(dim1):
        (dim2,) = struct.unpack_from("<Q", body, pos)
        pos += 8
        for _ in range(dim2):
            mode, size = _header(body, pos)
            assert mode == 0
            pk, after = _ciphertext(body, pos + 16)
            assert after == pos + size
            assert pk["ntt"] == 1 and pk["limbs"] == 3 and pk["size"] == 2
            filled.setdefault(idx, []).append(pk)
            pos = after
    assert pos == len(body) and list(filled) == [1] and len(filled[1]) == 2
    # 4: a secret key = plaintext of K * N coefficients
    body, off = _object(buf, off)
    assert struct.unpack_from("<4Q", body, 0) == _parms_id(1, 64, primes, 65537)
    count, scale = struct.unpack_from("<Qd", body, 32)
    data, end = _array(body, 48)
    assert count == 192 and scale == 1.0 and data == list(range(1000, 1192)) and end == len(body)
    assert off == len(buf)


@pytest.mark.gpu
def test_keys_and_ciphertexts_travel_between_factories():
    from abc_amd import capi
    assert os.path.exists(capi.LIB_PATH), "libabc_hip.so missing: python -m abc_amd.build"
    subprocess.check_call(["make", "-C", RT, "test_seal_wire"], stdout=subprocess.DEVNULL)
    p = subprocess.run([os.path.join(RT, "test_seal_wire")], capture_output=True, text=True, timeout=600)
    print(p.stdout[-3000:])
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert " 0 failed" in p.stdout
