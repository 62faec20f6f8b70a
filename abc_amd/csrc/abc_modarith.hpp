// abc_modarith.hpp -- 64-bit modular arithmetic for gfx950 (CDNA4) device code.
//
// All RNS limbs are unsigned 64-bit residues of primes up to 61 bits (SEAL's user primes are
// <= 60 bits, its BEHZ auxiliary primes 61 bits; see include/abc_hip.h).  CDNA4 has no 64x64->128
// multiplier: every wide product is built from v_mad_u64_u32 / v_mul_hi_u32, so the routines below
// are written to minimise the number of 32x32 multiplies:
//   * mul_shoup_lazy : constant operand with precomputed quotient  -> mulhi64 + 2 mullo64
//   * barrett_reduce : 128-bit value  -> one mulhi64 + one mullo64 (mu pre-shifted so the estimate
//                      is exactly the high word: mu = floor(2^(k+63)/q), k = bitlen(q))
// Every routine returns fully reduced residues where the name does not say "lazy", so results are
// canonical and bit-comparable with any other correct implementation (oracle/, SEAL).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace abc {

typedef uint64_t u64;  // same type as the C ABI's uint64_t
typedef uint32_t u32;
struct alignas(16) u64x2 {
  u64 x, y;
};

// Per-modulus constants, resident in device memory (one entry per prime used by a context).
struct Mod {
  u64 q;       // modulus
  u64 mu;      // floor(2^(k+63) / q), k = bitlen(q)   (q = 2^32 (m_tilde) is special-cased by callers)
  u64 two_q;   // 2q
  u32 shift;   // k - 1
  u32 bits;    // k
  u64 inv_n;   // N^-1 mod q            (inverse NTT scaling)
  u64 inv_n_s; // Shoup quotient of inv_n
  double qd;   // (double) q            (fp64 fast path, primes < 2^50 only)
  double qinv; // 1.0 / q
  double inv_n_c;   // N^-1 mod q centred into (-q/2, q/2]   (fp64 path)
  double inv_n_cq;  // inv_n_c / q
};

struct U128 {
  u64 lo, hi;
};

__device__ __forceinline__ u64 mulhi64(u64 a, u64 b) {
  return (u64)__umul64hi((unsigned long long)a, (unsigned long long)b);
}

__device__ __forceinline__ U128 mul_wide(u64 a, u64 b) {
  U128 r;
  r.lo = a * b;
  r.hi = mulhi64(a, b);
  return r;
}

__device__ __forceinline__ void add128(U128 &acc, const U128 &x) {
  u64 lo = acc.lo + x.lo;
  acc.hi += x.hi + (lo < acc.lo ? 1ull : 0ull);
  acc.lo = lo;
}

// acc += a*b  (128-bit lazy accumulation; caller bounds the number of summands, see barrett_reduce)
__device__ __forceinline__ void mac128(U128 &acc, u64 a, u64 b) { add128(acc, mul_wide(a, b)); }

// Reduce a 128-bit value x < 2^(k+63) to [0,q).  For q < 2^61 this admits at least 4 products of
// reduced operands (x < 4q^2), for 50-bit primes 2^13 of them.
__device__ __forceinline__ u64 barrett_reduce(const U128 &x, const Mod &m) {
  // q1 = floor(x / 2^(k-1)) fits 64 bits by the precondition
  u32 s = m.shift;
  u64 q1 = (x.lo >> s) | (x.hi << (64 - s));  // s in [19,60]: never 0 or 64 for supported primes
  u64 qh = mulhi64(q1, m.mu);
  u64 r = x.lo - qh * m.q;  // true remainder + {0,1,2}*q, fits 64 bits
  if (r >= m.two_q) r -= m.two_q;
  if (r >= m.q) r -= m.q;
  return r;
}

// Reduce an arbitrary 64-bit value to [0,q)
__device__ __forceinline__ u64 reduce64(u64 x, const Mod &m) {
  u64 q1 = x >> m.shift;
  u64 qh = mulhi64(q1, m.mu);
  u64 r = x - qh * m.q;
  if (r >= m.two_q) r -= m.two_q;
  if (r >= m.q) r -= m.q;
  return r;
}

__device__ __forceinline__ u64 mul_mod(u64 a, u64 b, const Mod &m) { return barrett_reduce(mul_wide(a, b), m); }

__device__ __forceinline__ u64 add_mod(u64 a, u64 b, u64 q) {
  u64 s = a + b;
  return s >= q ? s - q : s;
}
__device__ __forceinline__ u64 sub_mod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }
__device__ __forceinline__ u64 neg_mod(u64 a, u64 q) { return a ? q - a : 0; }

// x - c if that does not borrow, else x  (sub + carry-select: no separate 64-bit compare)
__device__ __forceinline__ u64 csub(u64 x, u64 c) {
  unsigned long t;
  const bool borrow = __builtin_usubl_overflow((unsigned long)x, (unsigned long)c, &t);
  return borrow ? x : (u64)t;
}

// Shoup multiplication by a constant w with ws = floor(w * 2^64 / q): result in [0, 2q) for ANY y.
// y*w - h*q is evaluated as y*w + h*(2^64 - q) so that both low products chain through v_mad_u64_u32's
// 64-bit addend (2 mad + 4 mul_lo + 2 add3) instead of two products and a 64-bit subtract.
__device__ __forceinline__ u64 mul_shoup_lazy(u64 y, u64 w, u64 ws, u64 q) {
  const u64 h = mulhi64(y, ws);
  const u64 nq = 0 - q;
  const u32 y0 = (u32)y, y1 = (u32)(y >> 32), w0 = (u32)w, w1 = (u32)(w >> 32);
  const u32 h0 = (u32)h, h1 = (u32)(h >> 32), n0 = (u32)nq, n1 = (u32)(nq >> 32);
  u64 acc = (u64)y0 * w0;
  acc = (u64)h0 * n0 + acc;
  const u32 hi = (u32)(acc >> 32) + y0 * w1 + y1 * w0 + h0 * n1 + h1 * n0;
  return ((u64)hi << 32) | (u32)acc;
}
// Same with a cheaper quotient estimate: the y0*ws0 partial product and the carries out of the two middle
// partial products are dropped, so the estimate is short by at most 2 and the result lies in [0, 4q).  One
// v_mad_u64_u32 + two v_mul_hi_u32 instead of one v_mul_hi_u32 + three v_mad_u64_u32 with 64-bit addends.
__device__ __forceinline__ u64 mul_shoup_lazy4(u64 y, u64 w, u64 ws, u64 q) {
  const u32 y0 = (u32)y, y1 = (u32)(y >> 32), s0 = (u32)ws, s1 = (u32)(ws >> 32);
  const u64 h = (u64)y1 * s1 + (u64)__umulhi(y1, s0) + (u64)__umulhi(y0, s1);
  const u64 nq = 0 - q;
  const u32 w0 = (u32)w, w1 = (u32)(w >> 32);
  const u32 h0 = (u32)h, h1 = (u32)(h >> 32), n0 = (u32)nq, n1 = (u32)(nq >> 32);
  u64 acc = (u64)y0 * w0;
  acc = (u64)h0 * n0 + acc;
  const u32 hi = (u32)(acc >> 32) + y0 * w1 + y1 * w0 + h0 * n1 + h1 * n0;
  return ((u64)hi << 32) | (u32)acc;
}
__device__ __forceinline__ u64 mul_shoup(u64 y, u64 w, u64 ws, u64 q) {
  u64 r = mul_shoup_lazy(y, w, ws, q);
  return r >= q ? r - q : r;
}

__device__ __forceinline__ u32 bitrev32(u32 x, int bits) { return __brev(x) >> (32 - bits); }
// Galois automorphism x -> x^elt on an NTT-form limb (SEAL bit-reversed order) as a gather: slot i holds the evaluation
// at psi^(2 bitrev(i) + 1) and receives the slot holding that exponent times elt.  The map sends every aligned block of
// 2^k slots onto an aligned block of 2^k slots (the high bits of the source depend only on the high bits of i), so a
// wavefront that reads 64 consecutive slots gathers from one 512-byte segment.  GAL = false: identity, decided at
// compile time (a run-time test per element would turn the callers' batched loads into a branch per word).
template <bool GAL>
__device__ __forceinline__ u32 galois_ntt_src(u32 i, u32 elt, int logn) {
  if (!GAL) return i;
  const u32 rev = bitrev32(i + (1u << logn), logn + 1);
  const u32 idx = (u32)((((u64)elt * rev) >> 1) & (u64)((1u << logn) - 1));
  return bitrev32(idx, logn);
}

// Galois automorphism x -> x^elt on a COEFFICIENT-form limb as a gather: coefficient j of the result is +-coefficient i of the
// operand with i * elt = j or j + N (mod 2N); with einv = elt^-1 mod 2N: i0 = j * einv mod 2N, i = i0 mod N, negated iff i0 >= N.
__device__ __forceinline__ u32 galois_coef_src(u32 j, u32 einv, int logn, bool &neg) {
  const u32 i0 = (j * einv) & ((2u << logn) - 1u);
  neg = (i0 >> logn) & 1u;
  return i0 & ((1u << logn) - 1u);
}

}  // namespace abc
