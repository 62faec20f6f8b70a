// CkksEncoder -- host-side CKKS canonical-embedding encoder / decoder for the plugin classes (C++ twin of
// abc_amd/ckks_encoder.py; checked against the CPU reference encoder in tests/cpp/test_hip_ckks_runtime.cpp).
//
// The reference has no CKKS code (src/runtime/SealCiphertextFactory.cpp:74 hard-codes BFV; CMakeLists.txt:216 leaves the
// HAVE_SEAL_CKKS hook), so this follows the published encoding with SEAL's slot order: slot i <-> evaluation at
// zeta^(3^i), zeta = exp(i pi / N), conjugates at zeta^(-3^i).  It produces / consumes COEFFICIENT-form residues [nl][N];
// the device turns them into the NTT form ciphertexts and plaintexts travel in (abc_hip_ntt_limbs).  Floating point: off
// the hot path, agrees with any other CKKS encoder to rounding error, not bit for bit.
#pragma once

#include <cmath>
#include <complex>
#include <cstdint>
#include <stdexcept>
#include <vector>

class CkksEncoder {
  size_t n = 0;
  std::vector<uint64_t> primes;            // data limbs q_0 .. q_{L-1}
  std::vector<uint32_t> idx, idxConj;      // slot -> position of its evaluation point
  std::vector<std::complex<double>> twist;  // exp(-i pi k / N)
  // Garner constants for every prefix of the chain: inv[i][j] = q_j^-1 mod q_i (j < i)
  std::vector<std::vector<uint64_t>> inv;

  typedef unsigned __int128 u128;
  static uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)((u128)a * b % q); }
  static uint64_t powmod(uint64_t b, uint64_t e, uint64_t q) {
    uint64_t r = 1 % q;
    b %= q;
    while (e) {
      if (e & 1) r = mulmod(r, b, q);
      b = mulmod(b, b, q);
      e >>= 1;
    }
    return r;
  }
  // in-place iterative radix-2 FFT, numpy convention: forward X[k] = sum x[m] exp(-2 pi i k m / n)
  static void fft(std::vector<std::complex<double>> &a, bool inverse) {
    const size_t len = a.size();
    for (size_t i = 1, j = 0; i < len; i++) {
      size_t bit = len >> 1;
      for (; j & bit; bit >>= 1) j ^= bit;
      j ^= bit;
      if (i < j) std::swap(a[i], a[j]);
    }
    for (size_t l = 2; l <= len; l <<= 1) {
      const double ang = 2 * M_PI / (double)l * (inverse ? 1 : -1);
      const std::complex<double> wl(std::cos(ang), std::sin(ang));
      for (size_t i = 0; i < len; i += l) {
        std::complex<double> w(1);
        for (size_t k = 0; k < l / 2; k++) {
          // recompute the twiddle from the angle every 64 steps: keeps the error of the running product at 1e-15
          if ((k & 63) == 0) w = std::complex<double>(std::cos(ang * (double)k), std::sin(ang * (double)k));
          const std::complex<double> u = a[i + k], v = a[i + k + l / 2] * w;
          a[i + k] = u + v;
          a[i + k + l / 2] = u - v;
          w *= wl;
        }
      }
    }
    if (inverse)
      for (auto &x : a) x /= (double)len;
  }

 public:
  CkksEncoder() = default;
  CkksEncoder(size_t ringDegree, const std::vector<uint64_t> &dataPrimes) : n(ringDegree), primes(dataPrimes) {
    const size_t m2 = 2 * n;
    uint64_t g = 1;
    for (size_t i = 0; i < n / 2; i++) {
      idx.push_back((uint32_t)((g - 1) >> 1));
      idxConj.push_back((uint32_t)((m2 - g - 1) >> 1));
      g = (g * 3) % m2;
    }
    for (size_t k = 0; k < n; k++) twist.emplace_back(std::cos(M_PI * (double)k / (double)n), -std::sin(M_PI * (double)k / (double)n));
    inv.resize(primes.size());
    for (size_t i = 0; i < primes.size(); i++)
      for (size_t j = 0; j < i; j++) inv[i].push_back(powmod(primes[j] % primes[i], primes[i] - 2, primes[i]));
  }
  size_t slots() const { return n / 2; }

  // real vector (<= N/2 slots, the rest zero) -> residues [nl][N], coefficient form
  void encode(const std::vector<double> &values, double scale, int nl, uint64_t *out) const {
    if (values.size() > n / 2) throw std::runtime_error("CKKS encode: more values than slots");
    std::vector<std::complex<double>> w(n);
    for (size_t i = 0; i < values.size(); i++) {
      w[idx[i]] = values[i];
      w[idxConj[i]] = values[i];
    }
    fft(w, false);
    for (size_t k = 0; k < n; k++) {
      const double c = (w[k] * twist[k]).real() / (double)n * scale;
      if (!(std::fabs(c) < 4.6e18)) throw std::runtime_error("CKKS encode: scale too large for 64-bit coefficient rounding");
      const int64_t r = (int64_t)std::llrint(c);
      for (int j = 0; j < nl; j++) {
        const int64_t q = (int64_t)primes[j];
        int64_t v = r % q;
        if (v < 0) v += q;
        out[(size_t)j * n + k] = (uint64_t)v;
      }
    }
  }

  // residues [nl][N] in coefficient form -> the N/2 slot values (real parts)
  void decode(const uint64_t *res, int nl, double scale, std::vector<double> &values) const {
    // Q/2 and Q as little-endian multi-word integers
    std::vector<uint64_t> Q(nl + 1, 0);
    Q[0] = 1;
    for (int i = 0; i < nl; i++) {
      u128 carry = 0;
      for (auto &wd : Q) {
        const u128 t = (u128)wd * primes[i] + carry;
        wd = (uint64_t)t;
        carry = t >> 64;
      }
    }
    std::vector<uint64_t> halfQ(Q);
    for (int wd = 0; wd <= nl; wd++) halfQ[wd] = (Q[wd] >> 1) | (wd < nl ? Q[wd + 1] << 63 : 0);
    std::vector<std::complex<double>> w(n);
    std::vector<uint64_t> d(nl), acc(nl + 1);
    for (size_t k = 0; k < n; k++) {
      // Garner: mixed-radix digits of the residue vector, x = d0 + q0 (d1 + q1 (d2 + ...))
      for (int i = 0; i < nl; i++) {
        const uint64_t q = primes[i];
        uint64_t t = res[(size_t)i * n + k] % q;
        for (int j = 0; j < i; j++) {
          const uint64_t dj = d[j] % q;
          t = mulmod(t >= dj ? t - dj : t + q - dj, inv[i][j], q);
        }
        d[i] = t;
      }
      std::fill(acc.begin(), acc.end(), 0);
      acc[0] = d[nl - 1];
      for (int i = nl - 2; i >= 0; i--) {
        u128 carry = d[i];
        for (auto &wd : acc) {
          const u128 t = (u128)wd * primes[i] + carry;
          wd = (uint64_t)t;
          carry = t >> 64;
        }
      }
      bool neg = false;
      for (int wd = nl; wd >= 0; wd--)
        if (acc[wd] != halfQ[wd]) { neg = acc[wd] > halfQ[wd]; break; }
      if (neg) {  // acc = Q - acc
        unsigned char borrow = 0;
        for (int wd = 0; wd <= nl; wd++) {
          const u128 t = (u128)Q[wd] - acc[wd] - borrow;
          acc[wd] = (uint64_t)t;
          borrow = (unsigned char)((t >> 64) & 1);
        }
      }
      double mag = 0;
      for (int wd = nl; wd >= 0; wd--) mag = mag * 18446744073709551616.0 + (double)acc[wd];
      w[k] = std::conj(twist[k]) * ((neg ? -mag : mag) / scale);
    }
    fft(w, true);
    values.resize(n / 2);
    for (size_t i = 0; i < n / 2; i++) values[i] = w[idx[i]].real() * (double)n;
  }
};
