// abc_ntt.hpp -- LDS-staged negacyclic NTT / INTT building blocks for gfx950.
//
// One workgroup transforms one contiguous block of 2^LB coefficients (LB = 10..14: 8..128 KiB of the
// CU's 160 KiB LDS) of one RNS limb.  Each thread keeps 16 coefficients in VGPRs and performs up to
// four butterfly stages (radix-16) between two LDS exchanges, so a 2^14-point transform touches LDS
// three times and HBM exactly once in and once out.  Ordering follows the reference's SEAL
// dependency: forward = Cooley-Tukey, natural in -> bit-reversed out; inverse = Gentleman-Sande,
// bit-reversed in -> natural out, with twiddles tw[m+i] = psi^bitrev(m+i) (psi = minimal primitive
// 2N-th root) -- the layout every NTT-form ciphertext, key and plaintext of the reference's SEAL
// runtime uses (call sites: src/runtime/SealCiphertext.cpp:104-105,122-123,159,196).
// Butterflies are Harvey lazy ([0,4q) forward, [0,2q) inverse); callers canonicalise in the store
// functor, so outputs are bit-exact with the oracle.
//
// A block may be a sub-transform of a larger N (N = 2^15, 2^16 do not fit LDS): the caller passes the
// number of stages already done (S0) and the block index b, and the first S0 stages are done by the
// strided global pass in abc_ntt.hip.
#pragma once

#include "abc_modarith.hpp"

namespace abc {

// LDS padding: 4 extra words every 64 keeps the stride-2^k accesses of the middle passes
// conflict-free for ds_read_b64/ds_write_b64 (bank = (addr/4) mod 64).
__device__ __forceinline__ int lds_pad(int i) { return i + ((i >> 6) << 2); }
constexpr int lds_words(int lb) { return (1 << lb) + ((1 << lb) >> 6) * 4; }

// Twiddle tables are interleaved {w, floor(w * 2^64 / q)} pairs so that one 16-byte load (s_load_dwordx4 where the
// index is wave-uniform, global_load_dwordx4 otherwise) fetches everything a butterfly needs.
struct NttTable {
  const u64x2 *tw;   // [N] forward twiddles {w, Shoup quotient}, bit-reversed order: tw[m+i].x = psi^bitrev(m+i)
  const u64x2 *itw;  // [N] inverse twiddles (itw[m+i].x = tw[m+i].x^-1)
};
// fp64 twin of the table for primes below 2^50: {w centred into (-q/2, q/2], w / q}
struct alignas(16) f64x2 {
  double x, y;
};
// The tables are read through the constant address space: they never change while a kernel runs, and saying so lets
// the compiler keep wave-uniform twiddles on the scalar unit (s_load, lgkmcnt) even in kernels that store to HBM
// between two transforms.  On gfx950 loads and stores share one in-order vmcnt, so a twiddle fetched with a vector
// load would make the wave wait for every store issued before it -- which is exactly what must not happen if the
// stores of one transform are to drain under the first passes of the next.
#define ABC_CONST_AS __attribute__((address_space(4)))
struct FpTable {
  const ABC_CONST_AS f64x2 *tw, *itw;
};
__device__ __forceinline__ f64x2 tw_load(const ABC_CONST_AS f64x2 *p) {
  f64x2 r;
  r.x = p->x;
  r.y = p->y;
  return r;
}
__device__ __forceinline__ u64x2 tw_load(const u64x2 *p) { return *p; }

// ---- fp64 residue arithmetic (primes < 2^50) ------------------------------------------------------------
// Residues are integer-valued doubles, signed and lazily reduced.  Every operation below is exact as long as
// the magnitudes stay below 2^53, so results are the same integers mod q that the 64-bit path produces; only the
// quotient estimate c is approximate, which changes WHICH representative comes out, never its residue class.
//   v = y*w - c*q,  c = rint(y * (w/q)):  h = fl(y*w), l = y*w - h (FMA, exact), d = h - c*q (FMA, exact because
//   |d| < 2^53 and d is a multiple of ulp(h) or an integer), v = d + l.
//   |c - y*w/q| <= 1/2 + |y| * |w/q| * 2^-52  and |w| <= q/2, so  |v| <= q/2 + |y| * q * 2^-53:
//   one stage grows a bound Y on |value| to at most Y (1 + q 2^-53) + q/2.
// Primes of 49 and 50 bits (q 2^-53 >= 1/16) are re-centred at the start of every register pass (FpK::red),
// smaller ones need no reduction inside a forward transform (2^50 (1 + 2^-5)^14 + 7.7 q < 2^52).
// 8 DP instructions per butterfly instead of 16 integer ones, no carry chains (no VCC hazards), measured
// 3.2 T butterflies/s against 2.2 T/s in isolation (tools/microbench.py probes 200/202).
struct FpK {
  double q, qinv;
  bool red;  // 49- and 50-bit primes
};
#pragma clang fp contract(off)
__device__ __forceinline__ double fp_mul_lazy(double y, double w, double wq, double q) {
  const double h = y * w;
  const double l = __builtin_fma(y, w, -h);
  const double c = __builtin_rint(y * wq);
  return __builtin_fma(-c, q, h) + l;
}
// centred representative: |result| <= q/2 (+ |x| q 2^-105, nothing)
__device__ __forceinline__ double fp_centre(double x, double q, double qinv) {
  return __builtin_fma(-__builtin_rint(x * qinv), q, x);
}
// exact conversions: u64 below 2^52 <-> integer-valued double
__device__ __forceinline__ double fp_from_u64(u64 v) {
  return __longlong_as_double((long long)(v | 0x4330000000000000ull)) - 4503599627370496.0;
}
// any lazy value -> canonical [0, q) as u64: centre, add q where negative (sign mask, no VCC), strip the exponent
__device__ __forceinline__ u64 fp_to_canon(double x, double q, double qinv) {
  double r = fp_centre(x, q, qinv);
  const u32 neg = (u32)((int)(u32)((u64)__double_as_longlong(r) >> 32) >> 31);
  const u64 qb = (u64)__double_as_longlong(q);
  const u64 add = ((u64)((u32)(qb >> 32) & neg) << 32) | (u64)((u32)qb & neg);
  r += __longlong_as_double((long long)add);
  return (u64)__double_as_longlong(r + 4503599627370496.0) & 0x000fffffffffffffull;
}
// any lazy value -> a non-negative representative in [q/2, 3q/2] as u64 (for consumers that reduce anyway)
__device__ __forceinline__ u64 fp_to_lazy(double x, double q, double qinv) {
  const double r = fp_centre(x, q, qinv);
  return (u64)__double_as_longlong(r + (4503599627370496.0 + q)) & 0x000fffffffffffffull;
}
__host__ __device__ __forceinline__ bool fp_ok(u32 bits) { return bits <= 50; }

// ---- packed half-done limbs (scratch traffic between the kernels of a split sequence) --------------------------------------
// A half-done limb is written once and read once; as raw doubles it costs 8 bytes per coefficient each way although a centred
// residue modulo a 40-bit prime is a signed 40-bit integer.  Packed form of a limb of N coefficients (inside the limb's
// ordinary 8N-byte scratch slot): a plane of N low dwords, then a plane of N high bytes (kind 1: primes of at most 40 bits,
// 5 bytes per coefficient) or N high half-words (kind 2: at most 48 bits, 6 bytes).  Planes, not interleaved records: a
// wavefront's store instruction then still covers whole 32-byte sectors (64 consecutive dwords / bytes), and the consumer's
// coefficient pair is one 8-byte plus one 2- or 4-byte load.  Kind 0: raw doubles (49- and 50-bit primes).
// The value is the CENTRED representative c, |c| <= q/2, stored as a two's-complement integer: bits(c + 1.5 * 2^52) carries c's
// low 48 bits in its mantissa; the reader rebuilds those bits with the sign-extended high part and subtracts the constant.
__host__ __device__ __forceinline__ int pack_kind(u32 bits) { return bits <= 40 ? 1 : bits <= 48 ? 2 : 0; }
constexpr double kPackMagic = 6755399441055744.0;  // 2^52 + 2^51
template <int PK>
__device__ __forceinline__ void pack_store(double *limb, size_t n, size_t idx, double centred) {
  const u64 b = (u64)__double_as_longlong(centred + kPackMagic);
  reinterpret_cast<u32 *>(limb)[idx] = (u32)b;
  if (PK == 1) reinterpret_cast<unsigned char *>(limb)[4 * n + idx] = (unsigned char)(b >> 32);
  else reinterpret_cast<unsigned short *>(reinterpret_cast<unsigned char *>(limb) + 4 * n)[idx] = (unsigned short)(b >> 32);
}
// raw words of the coefficient pair (idx, idx + 1), idx even: .x = the two low dwords, .y = the two high parts
template <int PK>
__device__ __forceinline__ u64x2 pack_load_pair(const double *limb, size_t n, size_t idx) {
  u64x2 r;
  r.x = *reinterpret_cast<const u64 *>(reinterpret_cast<const u32 *>(limb) + idx);
  if (PK == 1) r.y = *reinterpret_cast<const unsigned short *>(reinterpret_cast<const unsigned char *>(limb) + 4 * n + idx);
  else r.y = *reinterpret_cast<const u32 *>(reinterpret_cast<const unsigned char *>(limb) + 4 * n + 2 * idx);
  return r;
}
template <int PK>
__device__ __forceinline__ void pack_decode_pair(const u64x2 &r, double &v0, double &v1) {
  const u32 h = (u32)r.y;
  const int h0 = PK == 1 ? (int)(signed char)(h & 0xffu) : (int)(short)(h & 0xffffu);
  const int h1 = PK == 1 ? (int)(signed char)((h >> 8) & 0xffu) : ((int)h >> 16);
  const u64 b0 = ((u64)(u32)(0x43380000 + h0) << 32) | (u32)r.x;
  const u64 b1 = ((u64)(u32)(0x43380000 + h1) << 32) | (u32)(r.x >> 32);
  v0 = __longlong_as_double((long long)b0) - kPackMagic;
  v1 = __longlong_as_double((long long)b1) - kPackMagic;
}

// ---- arithmetic policies of the register passes ---------------------------------------------------------
// Integer (Harvey lazy) butterflies.
// GUARD = false: no per-stage correction of X and the cheaper quotient estimate (mul_shoup_lazy4, product in
// [0,4q)); every stage then adds at most 4q to a value, so after all logN <= 14 stages of a block a canonical
// input stays below (4*14 + 1) q = 57q -- usable whenever that fits 64 bits (q <= 57 bits), which holds for every
// SEAL default prime up to N = 32768 and for the CKKS chains here; the 61-bit BEHZ primes (and 58..60-bit user
// primes) take the guarded form.  Inverse: inputs/outputs in [0,2q).
template <bool GUARD>
struct IntArith {
  using E = u64;
  using TW = u64x2;
  using Table = NttTable;
  struct K {
    u64 q, two_q;
  };
  __device__ __forceinline__ static K consts(const Mod &m) { return K{m.q, m.two_q}; }
  __device__ __forceinline__ static void fwd(E &X, E &Y, const TW tp, const K &k) {
    if (GUARD) {
      const u64 a = csub(X, k.two_q);
      const u64 v = mul_shoup_lazy(Y, tp.x, tp.y, k.q);
      X = a + v;
      Y = a + k.two_q - v;
    } else {
      const u64 a = X;
      const u64 v = mul_shoup_lazy4(Y, tp.x, tp.y, k.q);
      X = a + v;
      Y = a + (k.two_q << 1) - v;
    }
  }
  __device__ __forceinline__ static void inv(E &X, E &Y, const TW tp, const K &k) {
    const u64 a = X, c = Y;
    X = csub(a + c, k.two_q);
    Y = mul_shoup_lazy(a + k.two_q - c, tp.x, tp.y, k.q);
  }
  template <int PASS> __device__ __forceinline__ static void fwd_begin(E (&)[16], const K &) {}
  template <int PASS> __device__ __forceinline__ static void inv_begin(E (&)[16], const K &) {}
};

// fp64 butterflies.  Forward: X = a + v, Y = a - v.  Inverse: X = a + b (doubles every stage: 16x per pass, so
// every pass after the first starts by re-centring; the first one too for 49/50-bit primes, whose (a - b) may not
// exceed 8q), Y = (a - b) w.
struct FpArith {
  using E = double;
  using TW = f64x2;
  using Table = FpTable;
  using K = FpK;
  __device__ __forceinline__ static K consts(const Mod &m) { return K{m.qd, m.qinv, m.bits >= 49}; }
  __device__ __forceinline__ static void fwd(E &X, E &Y, const TW tp, const K &k) {
    const double a = X;
    const double v = fp_mul_lazy(Y, tp.x, tp.y, k.q);
    X = a + v;
    Y = a - v;
  }
  __device__ __forceinline__ static void inv(E &X, E &Y, const TW tp, const K &k) {
    const double a = X, c = Y;
    X = a + c;
    Y = fp_mul_lazy(a - c, tp.x, tp.y, k.q);
  }
  __device__ __forceinline__ static void centre16(E (&x)[16], const K &k) {
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = fp_centre(x[r], k.q, k.qinv);
  }
  template <int PASS> __device__ __forceinline__ static void fwd_begin(E (&x)[16], const K &k) {
    if (PASS > 0 && k.red) centre16(x, k);  // workgroup-uniform branch
  }
  template <int PASS> __device__ __forceinline__ static void inv_begin(E (&x)[16], const K &k) {
    if (PASS > 0 || k.red) centre16(x, k);
  }
};

// Forward BLOCK TAILS (at most 12 stages) that start from a CENTRED input, |x| <= q/2: the per-stage bound Y -> Y (1 + q 2^-53) + q/2
// gives, from 0.5 q,  1.06, 1.70, 2.41, 3.21, 4.11, 5.12, 6.26, 7.55 q after eight stages of a 50-bit prime (2^53 = 8 q * 2^50 / q:
// the ninth would pass it) and 7.6 q after ten, 9.6 q after twelve stages of a 49-bit one (limit 16 q).  So a tail needs ONE
// re-centring for 50-bit primes -- before pass 2, which every schedule reaches after at most eight stages (4+4, 3+4) -- and none for
// 49-bit ones, where FpArith re-centres before every pass (it serves whole 14-stage transforms from canonical inputs too).
// tests/test_fp64_exactness.py replays both schedules with Python integers.
struct FpTail : FpArith {
  template <int PASS> __device__ __forceinline__ static void fwd_begin(E (&x)[16], const K &k) {
    if (PASS == 2 && k.red && k.q >= 562949953421312.0) centre16(x, k);  // q >= 2^49: a 50-bit prime (workgroup-uniform)
  }
};

// ---- one radix-2^R register pass over NG = 16>>R groups -------------------------------------------
// Forward (CT): stages S..S+R-1 of the block-local transform.
template <class A, int LB, int S, int R>
__device__ __forceinline__ void fwd_pass(typename A::E (&x)[16], const int (&hi)[16 >> R], const typename A::Table &t,
                                         const typename A::K &kk, int S0, int b) {
  constexpr int NG = 16 >> R;
#pragma unroll
  for (int u = 0; u < R; u++) {
    const int half = 1 << (R - 1 - u);
#pragma unroll
    for (int g = 0; g < NG; g++) {
      // twiddle index of butterfly block: global stage S0+S+u
      int base = (1 << (S0 + S + u)) + (b << (S + u)) + (hi[g] << u);
#pragma unroll
      for (int k = 0; k < (1 << R); k++) {
        if (k & half) continue;
        int idx = base + (k >> (R - u));
        A::fwd(x[g * (1 << R) + k], x[g * (1 << R) + (k | half)], tw_load(t.tw + idx), kk);
      }
    }
  }
}

// The same pass with its twiddles in an LDS table of the BLOCK's own twiddles: entry 2^s + i = twiddle i of block-local stage s
// (global index (2^S0 + b) 2^s + i; block_twiddles_to_lds below fills it).  For the passes whose twiddle index differs from
// lane to lane: read through LDS they stay off the vector-memory counter, which on gfx950 is in order across loads AND
// stores -- a per-lane twiddle fetched by a vector load waits for every load issued before it, so operands of a LATER phase
// could never be requested ahead of the transform.
template <class A, int LB, int S, int R>
__device__ __forceinline__ void fwd_pass_lds(typename A::E (&x)[16], const int (&hi)[16 >> R], const typename A::TW *ltw,
                                             const typename A::K &kk) {
  constexpr int NG = 16 >> R;
#pragma unroll
  for (int u = 0; u < R; u++) {
    const int half = 1 << (R - 1 - u);
#pragma unroll
    for (int g = 0; g < NG; g++) {
      int base = (1 << (S + u)) + (hi[g] << u);
#pragma unroll
      for (int k = 0; k < (1 << R); k++) {
        if (k & half) continue;
        int idx = base + (k >> (R - u));
        A::fwd(x[g * (1 << R) + k], x[g * (1 << R) + (k | half)], ltw[idx], kk);
      }
    }
  }
}
// workgroup-cooperative fill of that table for the 2^LB-point block b behind S0 strided stages: values first (vector loads,
// issue early), then the LDS writes (call after whatever else should be requested first)
template <int LB, class TW, int PER>
__device__ __forceinline__ void block_twiddles_fetch(const ABC_CONST_AS TW *tw, int S0, int b, int tid, int nthreads, TW (&v)[PER]) {
#pragma unroll
  for (int r = 0; r < PER; r++) {
    int mm = tid + r * nthreads;
    if (mm < 1) mm = 1;
    if (mm > (1 << LB) - 1) mm = (1 << LB) - 1;
    const int sl = 31 - __builtin_clz(mm);
    v[r] = tw_load(tw + (((1 << S0) + b) << sl) + (mm - (1 << sl)));
  }
}
template <int LB, class TW, int PER>
__device__ __forceinline__ void block_twiddles_store(TW *ltw, int tid, int nthreads, const TW (&v)[PER]) {
#pragma unroll
  for (int r = 0; r < PER; r++) {
    const int mm = tid + r * nthreads;
    if (mm >= 1 && mm < (1 << LB)) ltw[mm] = v[r];
  }
}

// Inverse (GS): stages S+R-1 .. S (reverse order).
template <class A, int LB, int S, int R>
__device__ __forceinline__ void inv_pass(typename A::E (&x)[16], const int (&hi)[16 >> R], const typename A::Table &t,
                                         const typename A::K &kk, int S0, int b) {
  constexpr int NG = 16 >> R;
#pragma unroll
  for (int u = R - 1; u >= 0; u--) {
    const int half = 1 << (R - 1 - u);
#pragma unroll
    for (int g = 0; g < NG; g++) {
      int base = (1 << (S0 + S + u)) + (b << (S + u)) + (hi[g] << u);
#pragma unroll
      for (int k = 0; k < (1 << R); k++) {
        if (k & half) continue;
        int idx = base + (k >> (R - u));
        A::inv(x[g * (1 << R) + k], x[g * (1 << R) + (k | half)], tw_load(t.itw + idx), kk);
      }
    }
  }
}

template <class A, int LB, int S, int R>
__device__ __forceinline__ void inv_pass_lds(typename A::E (&x)[16], const int (&hi)[16 >> R], const typename A::TW *ltw,
                                             const typename A::K &kk) {
  constexpr int NG = 16 >> R;
#pragma unroll
  for (int u = R - 1; u >= 0; u--) {
    const int half = 1 << (R - 1 - u);
#pragma unroll
    for (int g = 0; g < NG; g++) {
      int base = (1 << (S + u)) + (hi[g] << u);
#pragma unroll
      for (int k = 0; k < (1 << R); k++) {
        if (k & half) continue;
        int idx = base + (k >> (R - u));
        A::inv(x[g * (1 << R) + k], x[g * (1 << R) + (k | half)], ltw[idx], kk);
      }
    }
  }
}

// ---- cross passes of the split transforms (abc_kernels_gsplit.hip, abc_kernels_bmul.hip) ----
// radix-2^R pass over the 2^R values one thread holds, one from each block (block index = array index): global stages 0..R-1
template <int R>
__device__ __forceinline__ void fwd_cross(double (&x)[1 << R], const FpTable &t, const FpK &kk) {
#pragma unroll
  for (int u = 0; u < R; u++) {
    const int half = 1 << (R - 1 - u);
#pragma unroll
    for (int k = 0; k < (1 << R); k++) {
      if (k & half) continue;
      FpArith::fwd(x[k], x[k | half], tw_load(t.tw + (1 << u) + (k >> (R - u))), kk);
    }
  }
}
template <int R>
__device__ __forceinline__ void inv_cross(double (&x)[1 << R], const FpTable &t, const FpK &kk) {
#pragma unroll
  for (int u = R - 1; u >= 0; u--) {
    const int half = 1 << (R - 1 - u);
#pragma unroll
    for (int k = 0; k < (1 << R); k++) {
      if (k & half) continue;
      FpArith::inv(x[k], x[k | half], tw_load(t.itw + (1 << u) + (k >> (R - u))), kk);
    }
    if (R > 4 && u == R - 4) {  // X = a + b doubles per stage: four stages on centred values stay below 2^53, a fifth needs this
#pragma unroll
      for (int k = 0; k < (1 << R); k++) x[k] = fp_centre(x[k], kk.q, kk.qinv);
    }
  }
}

// Index bookkeeping of one pass: group p has lo = p mod G, hi = p / G (G = 2^(LB-S-R)); its k-th element sits at
// (hi << (LB-S)) + (k << logG) + lo.  Group ownership is wave-contiguous: lane l of wave w owns groups
// w*64*NG + g*64 + l.  For the first pass (NG = 1) that is simply p = tid; for every later pass (S >= LB-10) a
// wave's 64*NG groups cover exactly the contiguous block of 1024 coefficients [1024 w, 1024 (w+1)), so those
// passes only touch LDS words written by the same wave and need no workgroup barrier (see wave_sync).
template <int LB, int S, int R>
struct PassIdx {
  static constexpr int T = (1 << LB) / 16;
  static constexpr int NG = 16 >> R;
  static constexpr int LOGG = LB - S - R;
  static constexpr int G = 1 << LOGG;
  static constexpr bool UNIFORM = (G >= 64);  // all 64 lanes of a wave share hi
  __device__ __forceinline__ static void groups(int tid, int (&hi)[NG], int (&lo)[NG]) {
    // the wave index as a scalar: where hi is wave-uniform (G >= 64) every twiddle index then stays in SGPRs
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
#pragma unroll
    for (int g = 0; g < NG; g++) {
      if (UNIFORM) {
        const int pw = wave * (64 * NG) + g * 64;  // scalar part of p; the lane part cannot reach bit LOGG
        hi[g] = pw >> LOGG;
        lo[g] = (pw + lane) & (G - 1);
      } else {
        const int p = wave * (64 * NG) + g * 64 + lane;
        hi[g] = p >> LOGG;
        lo[g] = p & (G - 1);
      }
    }
  }
  __device__ __forceinline__ static int elem(int hi, int lo, int k) { return (hi << (LB - S)) + (k << LOGG) + lo; }
};

template <int LB, int S, int R, class E>
__device__ __forceinline__ void lds_load(const E *lds, E (&x)[16], const int (&hi)[16 >> R], const int (&lo)[16 >> R]) {
  using P = PassIdx<LB, S, R>;
#pragma unroll
  for (int g = 0; g < P::NG; g++)
#pragma unroll
    for (int k = 0; k < (1 << R); k++) x[g * (1 << R) + k] = lds[lds_pad(P::elem(hi[g], lo[g], k))];
}
template <int LB, int S, int R, class E>
__device__ __forceinline__ void lds_store(E *lds, const E (&x)[16], const int (&hi)[16 >> R], const int (&lo)[16 >> R]) {
  using P = PassIdx<LB, S, R>;
#pragma unroll
  for (int g = 0; g < P::NG; g++)
#pragma unroll
    for (int k = 0; k < (1 << R); k++) lds[lds_pad(P::elem(hi[g], lo[g], k))] = x[g * (1 << R) + k];
}

// Exchange through LDS between two passes that are local to a wave: DS operations of one wave execute in
// issue order, so only the compiler must be kept from reordering them across this point.
// Both syncs fence the LDS ("local") address space only, so the compiler may move the global twiddle loads of the
// next pass above them and their latency hides under the current pass.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}
__device__ __forceinline__ void block_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Pass schedules (sum = LB, last forward pass has R = 2 so each lane ends with 32-byte contiguous runs).
template <int LB> struct Sched;
template <> struct Sched<10> { static constexpr int R0 = 4, R1 = 4, R2 = 2, R3 = 0; };
template <> struct Sched<11> { static constexpr int R0 = 4, R1 = 3, R2 = 2, R3 = 2; };
template <> struct Sched<12> { static constexpr int R0 = 4, R1 = 4, R2 = 2, R3 = 2; };
template <> struct Sched<13> { static constexpr int R0 = 4, R1 = 4, R2 = 3, R3 = 2; };
template <> struct Sched<14> { static constexpr int R0 = 4, R1 = 4, R2 = 4, R3 = 2; };

// ---- forward block transform -----------------------------------------------------------------------
// load(r, i)  -> u64 in [0,4q)  coefficient i (block-local natural index) for register slot r
// store(r, i, v)               v in [0,4q) lazily reduced value of output slot i (block-local,
//                              bit-reversed order) held in register slot r (r is a compile-time
//                              constant after unrolling, so callers may index register arrays with it)
// The final register layout is PassIdx<LB, LB-2, 2>: slot r = 4g+k holds element 4*(tid + T*g) + k.
template <int LB, class A, class Load, class Store, bool LTW = false>
__device__ __forceinline__ void ntt_fwd_block_a(typename A::E *lds, Load load, Store store, const typename A::Table &t,
                                                const Mod &m, int S0, int b, int tid_in = -1,
                                                const typename A::TW *ltw = nullptr /* LTW: block twiddle table in LDS */) {
  using SC = Sched<LB>;
  // a 1024-point block is one wavefront: callers may run several of them side by side in one workgroup (tid_in =
  // the lane id), so nothing in it may be a workgroup barrier
  const int tid = tid_in < 0 ? (int)threadIdx.x : tid_in;
  const typename A::K kk = A::consts(m);
  typename A::E x[16];
  {  // pass 0: global -> regs -> LDS
    constexpr int S = 0, R = SC::R0;
    using P = PassIdx<LB, S, R>;
    int hi[P::NG], lo[P::NG];
    P::groups(tid, hi, lo);
#pragma unroll
    for (int g = 0; g < P::NG; g++)
#pragma unroll
      for (int k = 0; k < (1 << R); k++) x[g * (1 << R) + k] = load(g * (1 << R) + k, P::elem(hi[g], lo[g], k));
    fwd_pass<A, LB, S, R>(x, hi, t, kk, S0, b);
    lds_store<LB, S, R>(lds, x, hi, lo);
  }
  if constexpr (LB <= 10) wave_sync(); else block_sync_lds();
  {  // pass 1
    constexpr int S = SC::R0, R = SC::R1;
    using P = PassIdx<LB, S, R>;
    int hi[P::NG], lo[P::NG];
    P::groups(tid, hi, lo);
    lds_load<LB, S, R>(lds, x, hi, lo);
    A::template fwd_begin<1>(x, kk);
    if constexpr (LTW && !P::UNIFORM) fwd_pass_lds<A, LB, S, R>(x, hi, ltw, kk);
    else fwd_pass<A, LB, S, R>(x, hi, t, kk, S0, b);
    if constexpr (SC::R3 == 0 && SC::R2 == 0) {
#pragma unroll
      for (int g = 0; g < P::NG; g++)
#pragma unroll
        for (int k = 0; k < (1 << R); k++) store(g * (1 << R) + k, P::elem(hi[g], lo[g], k), x[g * (1 << R) + k]);
      return;
    } else {
      lds_store<LB, S, R>(lds, x, hi, lo);  // in place: a thread rewrites exactly the words it read
    }
  }
  wave_sync();  // passes after the first are local to a wave (PassIdx): no workgroup barrier
  {  // pass 2
    constexpr int S = SC::R0 + SC::R1, R = SC::R2;
    using P = PassIdx<LB, S, R>;
    int hi[P::NG], lo[P::NG];
    P::groups(tid, hi, lo);
    lds_load<LB, S, R>(lds, x, hi, lo);
    A::template fwd_begin<2>(x, kk);
    if constexpr (LTW && !P::UNIFORM) fwd_pass_lds<A, LB, S, R>(x, hi, ltw, kk);
    else fwd_pass<A, LB, S, R>(x, hi, t, kk, S0, b);
    if constexpr (SC::R3 == 0) {
#pragma unroll
      for (int g = 0; g < P::NG; g++)
#pragma unroll
        for (int k = 0; k < (1 << R); k++) store(g * (1 << R) + k, P::elem(hi[g], lo[g], k), x[g * (1 << R) + k]);
      return;
    } else {
      lds_store<LB, S, R>(lds, x, hi, lo);  // in place: a thread rewrites exactly the words it read
    }
  }
  if constexpr (SC::R3 != 0) {
    wave_sync();
    constexpr int S = SC::R0 + SC::R1 + SC::R2, R = SC::R3;
    using P = PassIdx<LB, S, R>;
    int hi[P::NG], lo[P::NG];
    P::groups(tid, hi, lo);
    lds_load<LB, S, R>(lds, x, hi, lo);
    A::template fwd_begin<3>(x, kk);
    if constexpr (LTW && !P::UNIFORM) fwd_pass_lds<A, LB, S, R>(x, hi, ltw, kk);
    else fwd_pass<A, LB, S, R>(x, hi, t, kk, S0, b);
#pragma unroll
    for (int g = 0; g < P::NG; g++)
#pragma unroll
      for (int k = 0; k < (1 << R); k++) store(g * (1 << R) + k, P::elem(hi[g], lo[g], k), x[g * (1 << R) + k]);
  }
}


// ---- 1024-point forward block, one wavefront, 16-byte operand loads ----------------------------------------------------------
// Schedule 3 + 4 + 3 instead of 4 + 4 + 2: the first pass is two radix-8 groups on ADJACENT positions (lane l owns positions
// 2l and 2l + 1 of every 128-element row), so the caller fetches its sixteen operands as eight 16-byte loads, lane stride 16
// bytes -- fully coalesced -- where the 4 + 4 + 2 schedule needs sixteen 8-byte loads.  Same number of LDS exchanges (two).
//   xin slot g*8 + k  =  block element k*128 + 2*lane + g      (g = 0, 1;  k = 0..7)
// Per-lane twiddles (passes B and C) come from the LDS table `ltw` (block_twiddles_fetch/store); pass A's are wave-uniform.
template <class A, class Store>
__device__ __forceinline__ void ntt_fwd_tail1024_pairs(typename A::E *lds, typename A::E (&x)[16], Store store,
                                                       const typename A::Table &t, const Mod &m, int S0, int b, int lane,
                                                       const typename A::TW *ltw) {
  constexpr int LB = 10;
  const typename A::K kk = A::consts(m);
  {  // pass A: stages 0..2
    const int hi[2] = {0, 0};
    fwd_pass<A, LB, 0, 3>(x, hi, t, kk, S0, b);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      typename A::E *p = lds + lds_pad((k << 7) + 2 * lane);  // two adjacent words (an even index never straddles a pad)
      p[0] = x[k];
      p[1] = x[8 + k];
    }
  }
  wave_sync();
  {  // pass B: stages 3..6
    using P = PassIdx<LB, 3, 4>;
    int hi[P::NG], lo[P::NG];
    P::groups(lane, hi, lo);
    lds_load<LB, 3, 4>(lds, x, hi, lo);
    A::template fwd_begin<1>(x, kk);
    fwd_pass_lds<A, LB, 3, 4>(x, hi, ltw, kk);
    lds_store<LB, 3, 4>(lds, x, hi, lo);
  }
  wave_sync();
  {  // pass C: stages 7..9
    using P = PassIdx<LB, 7, 3>;
    int hi[P::NG], lo[P::NG];
    P::groups(lane, hi, lo);
    lds_load<LB, 7, 3>(lds, x, hi, lo);
    A::template fwd_begin<2>(x, kk);
    fwd_pass_lds<A, LB, 7, 3>(x, hi, ltw, kk);
#pragma unroll
    for (int g = 0; g < P::NG; g++)
#pragma unroll
      for (int k = 0; k < 8; k++) store(g * 8 + k, P::elem(hi[g], lo[g], k), x[g * 8 + k]);
  }
}

template <int LB, bool GUARD = true, class Load, class Store>
__device__ __forceinline__ void ntt_fwd_block(u64 *lds, Load load, Store store, const NttTable &t, const Mod &m, int S0,
                                              int b) {
  ntt_fwd_block_a<LB, IntArith<GUARD>>(lds, load, store, t, m, S0, b);
}

// ---- inverse block transform -----------------------------------------------------------------------
// load(r, i)  -> u64 in [0,2q)  input slot i (block-local, bit-reversed order); same register layout
//                              as the forward transform's final one
// store(r, i, v)               v in [0,2q): coefficient i BEFORE the N^-1 scaling (caller scales:
//                              for a sub-block the scaling belongs to the final strided pass)
template <int LB, class A, class Load, class Store, bool LTW = false>
__device__ __forceinline__ void ntt_inv_block_a(typename A::E *lds, Load load, Store store, const typename A::Table &t,
                                                const Mod &m, int S0, int b, int tid_in = -1,
                                                const typename A::TW *ltw = nullptr /* LTW: block table of INVERSE twiddles in LDS */) {
  using SC = Sched<LB>;
  // as in the forward transform: a 1024-point block is one wavefront (tid_in = the lane id), nothing in it may then be a
  // workgroup barrier
  const int tid = tid_in < 0 ? (int)threadIdx.x : tid_in;
  const typename A::K kk = A::consts(m);
  typename A::E x[16];
  constexpr int SA = SC::R0, SB = SC::R0 + SC::R1, SCc = SC::R0 + SC::R1 + SC::R2;
  if constexpr (SC::R3 != 0) {
    constexpr int S = SCc, R = SC::R3;
    using P = PassIdx<LB, S, R>;
    int hi[P::NG], lo[P::NG];
    P::groups(tid, hi, lo);
#pragma unroll
    for (int g = 0; g < P::NG; g++)
#pragma unroll
      for (int k = 0; k < (1 << R); k++) x[g * (1 << R) + k] = load(g * (1 << R) + k, P::elem(hi[g], lo[g], k));
    A::template inv_begin<0>(x, kk);
    if constexpr (LTW && !P::UNIFORM) inv_pass_lds<A, LB, S, R>(x, hi, ltw, kk);
    else inv_pass<A, LB, S, R>(x, hi, t, kk, S0, b);
    lds_store<LB, S, R>(lds, x, hi, lo);
    wave_sync();  // all passes but the last are local to a wave
  }
  {
    constexpr int S = SB, R = SC::R2;
    using P = PassIdx<LB, S, R>;
    int hi[P::NG], lo[P::NG];
    P::groups(tid, hi, lo);
    if constexpr (SC::R3 != 0) {
      lds_load<LB, S, R>(lds, x, hi, lo);
    } else {
#pragma unroll
      for (int g = 0; g < P::NG; g++)
#pragma unroll
        for (int k = 0; k < (1 << R); k++) x[g * (1 << R) + k] = load(g * (1 << R) + k, P::elem(hi[g], lo[g], k));
    }
    A::template inv_begin<(SC::R3 != 0 ? 1 : 0)>(x, kk);
    if constexpr (LTW && !P::UNIFORM) inv_pass_lds<A, LB, S, R>(x, hi, ltw, kk);
    else inv_pass<A, LB, S, R>(x, hi, t, kk, S0, b);
    lds_store<LB, S, R>(lds, x, hi, lo);
    wave_sync();
  }
  {
    constexpr int S = SA, R = SC::R1;
    using P = PassIdx<LB, S, R>;
    int hi[P::NG], lo[P::NG];
    P::groups(tid, hi, lo);
    lds_load<LB, S, R>(lds, x, hi, lo);
    A::template inv_begin<(SC::R3 != 0 ? 2 : 1)>(x, kk);
    if constexpr (LTW && !P::UNIFORM) inv_pass_lds<A, LB, S, R>(x, hi, ltw, kk);
    else inv_pass<A, LB, S, R>(x, hi, t, kk, S0, b);
    lds_store<LB, S, R>(lds, x, hi, lo);
    if constexpr (LB <= 10) wave_sync(); else block_sync_lds();
  }
  {
    constexpr int S = 0, R = SC::R0;
    using P = PassIdx<LB, S, R>;
    int hi[P::NG], lo[P::NG];
    P::groups(tid, hi, lo);
    lds_load<LB, S, R>(lds, x, hi, lo);
    A::template inv_begin<(SC::R3 != 0 ? 3 : 2)>(x, kk);
    if constexpr (LTW && !P::UNIFORM) inv_pass_lds<A, LB, S, R>(x, hi, ltw, kk);
    else inv_pass<A, LB, S, R>(x, hi, t, kk, S0, b);
#pragma unroll
    for (int g = 0; g < P::NG; g++)
#pragma unroll
      for (int k = 0; k < (1 << R); k++) store(g * (1 << R) + k, P::elem(hi[g], lo[g], k), x[g * (1 << R) + k]);
  }
}

template <int LB, class Load, class Store>
__device__ __forceinline__ void ntt_inv_block(u64 *lds, Load load, Store store, const NttTable &t, const Mod &m, int S0,
                                              int b) {
  ntt_inv_block_a<LB, IntArith<true>>(lds, load, store, t, m, S0, b);
}

// lazily reduced [0,4q) -> [0,q)
__device__ __forceinline__ u64 canon4(u64 v, const Mod &m) { return csub(csub(v, m.two_q), m.q); }
// output of an unguarded forward transform, [0,64q) -> [0,q)
__device__ __forceinline__ u64 canon32(u64 v, const Mod &m) {
  v = csub(v, m.two_q << 4);
  v = csub(v, m.two_q << 3);
  v = csub(v, m.two_q << 2);
  v = csub(v, m.two_q << 1);
  return canon4(v, m);
}
template <bool GUARD>
__device__ __forceinline__ u64 canon_fwd(u64 v, const Mod &m) { return GUARD ? canon4(v, m) : canon32(v, m); }
// a modulus may take the unguarded butterflies when 64 q <= 2^64
__host__ __device__ __forceinline__ bool unguarded_ok(u32 bits) { return bits <= 57; }
// scale by N^-1 and canonicalise (input [0,2q) or any 64-bit value)
__device__ __forceinline__ u64 scale_inv_n(u64 v, const Mod &m) { return mul_shoup(v, m.inv_n, m.inv_n_s, m.q); }

}  // namespace abc
