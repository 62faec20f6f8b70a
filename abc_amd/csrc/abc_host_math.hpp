// abc_host_math.hpp -- host-side number theory used once per context to build device tables.
//
// Product code (NOT the oracle): primes, primitive roots, inverses, Shoup quotients.  Mirrors what the
// reference obtains from SEAL when SealCiphertextFactory::setupSealContext builds its SEALContext
// (src/runtime/SealCiphertextFactory.cpp:72-100): CoeffModulus::BFVDefault (:80), PlainModulus::Batching
// (:83) and the NTT / BEHZ tables inside SEALContext (:86).
#pragma once

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <vector>

namespace abc {
namespace host {

typedef unsigned __int128 u128;

inline uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)((u128)a * b % q); }
inline uint64_t addmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128)a + b) % q); }
inline uint64_t submod(uint64_t a, uint64_t b, uint64_t q) { a %= q; b %= q; return a >= b ? a - b : a + q - b; }
inline uint64_t negmod(uint64_t a, uint64_t q) { a %= q; return a ? q - a : 0; }

inline uint64_t powmod(uint64_t b, uint64_t e, uint64_t q) {
  uint64_t r = 1 % q;
  b %= q;
  for (; e; e >>= 1) {
    if (e & 1) r = mulmod(r, b, q);
    b = mulmod(b, b, q);
  }
  return r;
}

// modular inverse by extended Euclid (modulus need not be prime: m_tilde = 2^32, 2N for Galois)
inline uint64_t invmod(uint64_t a, uint64_t q) {
  __int128 t0 = 0, t1 = 1, r0 = q, r1 = a % q;
  while (r1) {
    __int128 d = r0 / r1, tmp = t0 - d * t1;
    t0 = t1; t1 = tmp;
    tmp = r0 - d * r1; r0 = r1; r1 = tmp;
  }
  if (r0 != 1) throw std::runtime_error("abc: value not invertible");
  if (t0 < 0) t0 += q;
  return (uint64_t)t0;
}

inline bool is_prime(uint64_t n) {
  if (n < 2) return false;
  for (uint64_t p : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
    if (n == p) return true;
    if (n % p == 0) return false;
  }
  uint64_t d = n - 1;
  int s = 0;
  while (!(d & 1)) { d >>= 1; ++s; }
  for (uint64_t a : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
    uint64_t x = powmod(a, d, n);
    if (x == 1 || x == n - 1) continue;
    bool witness = true;
    for (int i = 1; i < s && witness; ++i) {
      x = mulmod(x, x, n);
      if (x == n - 1) witness = false;
    }
    if (witness) return false;
  }
  return true;
}

// `count` primes with exactly `bits` bits congruent to 1 mod 2*ntt_size, largest first.
inline std::vector<uint64_t> ntt_primes(size_t ntt_size, int bits, size_t count) {
  std::vector<uint64_t> out;
  uint64_t step = 2 * (uint64_t)ntt_size;
  uint64_t floor_ = 1ull << (bits - 1);
  for (uint64_t v = (1ull << bits) - step + 1; out.size() < count && v > floor_; v -= step)
    if (is_prime(v)) out.push_back(v);
  if (out.size() != count) throw std::runtime_error("abc: not enough NTT primes of the requested size");
  return out;
}

// smallest primitive 2N-th root of unity modulo prime q
inline uint64_t min_primitive_root(uint64_t two_n, uint64_t q) {
  if ((q - 1) % two_n) throw std::runtime_error("abc: prime is not NTT friendly");
  uint64_t cof = (q - 1) / two_n, g = 0;
  for (uint64_t c = 2; c < q && !g; ++c) {
    uint64_t cand = powmod(c, cof, q);
    if (powmod(cand, two_n >> 1, q) == q - 1) g = cand;
  }
  uint64_t sq = mulmod(g, g, q), best = g, cur = g;
  for (uint64_t i = 0; i < two_n / 2; ++i) {  // every odd power of g is a primitive 2N-th root
    if (cur < best) best = cur;
    cur = mulmod(cur, sq, q);
  }
  return best;
}

inline uint32_t bitrev(uint32_t x, int bits) {
  uint32_t r = 0;
  for (int i = 0; i < bits; ++i, x >>= 1) r = (r << 1) | (x & 1);
  return r;
}

inline uint64_t shoup(uint64_t w, uint64_t q) { return (uint64_t)(((u128)w << 64) / q); }

inline int bitlen(uint64_t v) {
  int b = 0;
  for (; v; v >>= 1) ++b;
  return b;
}

// product of a list of moduli, reduced mod p
inline uint64_t prod_mod(const std::vector<uint64_t> &ms, uint64_t p) {
  uint64_t v = 1 % p;
  for (uint64_t m : ms) v = mulmod(v, m % p, p);
  return v;
}

// bit length of the (multi-precision) product of moduli
inline int prod_bitlen(const std::vector<uint64_t> &ms) {
  std::vector<uint64_t> w(1, 1);
  for (uint64_t m : ms) {
    uint64_t carry = 0;
    for (auto &limb : w) {
      u128 p = (u128)limb * m + carry;
      limb = (uint64_t)p;
      carry = (uint64_t)(p >> 64);
    }
    if (carry) w.push_back(carry);
  }
  return (int)(w.size() - 1) * 64 + bitlen(w.back());
}

}  // namespace host
}  // namespace abc
