"""Host-side CKKS canonical-embedding encoder / decoder (numpy, double precision).

The reference has no CKKS code at all (ref:src/runtime/SealCiphertextFactory.cpp:74 hard-codes BFV), so
this follows the published CKKS encoding with SEAL's slot order: slot i <-> evaluation at zeta^(3^i), zeta = exp(i*pi/N),
conjugates at zeta^(-3^i).  It produces / consumes COEFFICIENT-form residues [nl][N]; the device turns them into the NTT form
ciphertexts and plaintexts travel in (abc_hip_ntt_limbs).  Floating point: values agree with any other CKKS encoder to
rounding error, not bit-for-bit.
"""
import numpy as np


def _slot_index(n):
    m2 = 2 * n
    g, idx, idx_conj = 1, [], []
    for _ in range(n // 2):
        idx.append((g - 1) >> 1)
        idx_conj.append((m2 - g - 1) >> 1)
        g = (g * 3) % m2
    return np.array(idx), np.array(idx_conj)


def encode(values, scale, n, primes):
    """complex/real vector (<= N/2 slots) -> uint64 residues [len(primes)][N], coefficient form."""
    z = np.zeros(n // 2, dtype=np.complex128)
    v = np.asarray(values, dtype=np.complex128)
    z[: len(v)] = v
    idx, idxc = _slot_index(n)
    w = np.zeros(n, dtype=np.complex128)
    w[idx] = z
    w[idxc] = np.conj(z)
    k = np.arange(n)
    coef = (np.fft.fft(w) * np.exp(-1j * np.pi * k / n)).real / n * scale
    if np.abs(coef).max() >= 2.0 ** 62:
        raise ValueError("scale too large for 64-bit coefficient rounding")
    r = np.rint(coef).astype(np.int64)
    out = np.empty((len(primes), n), dtype=np.uint64)
    for j, q in enumerate(primes):
        out[j] = np.mod(r, q).astype(np.uint64)  # python-style mod: result in [0, q)
    return out


def decode(residues, scale, n, primes):
    """uint64 residues [nl][N] in coefficient form -> complex vector of N/2 slots."""
    nl = residues.shape[0]
    qs = [int(q) for q in primes[:nl]]
    Q = 1
    for q in qs:
        Q *= q
    # CRT with python integers (exact), centred
    acc = np.zeros(n, dtype=object)
    for j, q in enumerate(qs):
        Qj = Q // q
        inv = pow(Qj % q, -1, q)
        acc = acc + np.array([int(x) for x in residues[j]], dtype=object) * (Qj * inv)
    acc = acc % Q
    centred = np.array([float(x - Q) if x > Q // 2 else float(x) for x in acc])
    k = np.arange(n)
    w = np.fft.ifft(centred / scale * np.exp(1j * np.pi * k / n)) * n
    idx, _ = _slot_index(n)
    return w[idx]
