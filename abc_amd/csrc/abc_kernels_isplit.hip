// abc_kernels_isplit.hip -- integer twins of the split key-switch kernels of abc_kernels_fused.hip ("split4" sequence), for
// CKKS chains at N = 2^14 that contain a prime above 2^50 (SEAL-typical {60,40,...,60} chains): same four launches, same
// buffers, same 84 limb transfers per multiply, 64-bit Harvey / Shoup / Barrett arithmetic (abc_modarith.hpp) instead of
// exact fp64.  Replaces, for those chains, the round-1 integer sequence (LDS-resident transform per (prime, limb) pair,
// streaming 128-bit inner product, LDS-resident mod-down: 104 limb transfers and twenty 139 KiB workgroups per multiply).
//   SealCiphertext::multiply / multiplyInplace (src/runtime/SealCiphertext.cpp:102-107,121-124) -> isplit_mul_relin
//   SealCiphertext::rotateRows (:52-61), relinearize                                          -> isplit_keyswitch
// Arithmetic notes: GUARD = a key prime above 57 bits (guarded butterflies, values < 4q); otherwise unguarded (values grow by
// 4q per stage, < 64q after the 14 stages of register pass + tail, canonicalised once).  Inner products accumulate 128-bit
// products of CANONICAL operands, four at a time (barrett_reduce admits 4 products of reduced operands for q < 2^61).
// Mixed chains: the arithmetic is a property of the prime a transform runs modulo, not of the chain.  In {60,40,40,40,60} only
// the transforms modulo the two 60-bit primes need integers; everything modulo a data prime below 2^50 -- 12 of the 20
// decomposition transforms, 6 of the 8 mod-down transforms and three quarters of the inner products -- takes the exact-fp64
// form of abc_kernels_fused.hip: `fpmask` (bit I = data prime I below 2^50) makes K1 and K2b write those half-done limbs as
// doubles, and the main step is launched twice, k_split4_main_fp over the fp64 primes and k_isplit_main over the others.
#include "abc_context.hpp"

namespace abc {

// x*y mod q for two residues |x|, |y| <= q in exact fp64 (as fp_mulmod of abc_kernels_fused.hip)
__device__ __forceinline__ double i_fp_mulmod(double x, double y, double q, double qinv) {
  const double h = x * y;
  const double l = __builtin_fma(x, y, -h);
  const double c = __builtin_rint(h * qinv);
  return __builtin_fma(-c, q, h) + l;
}

// radix-2^R pass over the 2^R values one thread holds, one from each 1024-point block (array index = block index): the first R
// stages of a forward transform / the last R of an inverse one, in the arithmetic A of the prime (integer or fp64)
template <int R, class A>
__device__ __forceinline__ void x_fwd_cross(typename A::E (&x)[1 << R], const typename A::Table &t, const typename A::K &kk) {
  static_assert(R <= 5, "at R = 6 the integer butterflies are left in a rolled stage loop and the array goes to scratch memory");
#pragma unroll
  for (int u = 0; u < R; u++) {
    const int half = 1 << (R - 1 - u);
#pragma unroll
    for (int k = 0; k < (1 << R); k++) {
      if (k & half) continue;
      A::fwd(x[k], x[k | half], tw_load(t.tw + (1 << u) + (k >> (R - u))), kk);
    }
  }
}
template <int R, class A>
__device__ __forceinline__ void x_inv_cross(typename A::E (&x)[1 << R], const typename A::Table &t, const typename A::K &kk) {
  static_assert(R <= 5, "see x_fwd_cross");
#pragma unroll
  for (int u = R - 1; u >= 0; u--) {
    const int half = 1 << (R - 1 - u);
#pragma unroll
    for (int k = 0; k < (1 << R); k++) {
      if (k & half) continue;
      A::inv(x[k], x[k | half], tw_load(t.itw + (1 << u) + (k >> (R - u))), kk);
    }
  }
}

template <int LB, class TW, int PER>
__device__ __forceinline__ void block_twiddles_fetch_g(const TW *tw, int S0, int b, int tid, int nthreads, TW (&v)[PER]) {
#pragma unroll
  for (int r = 0; r < PER; r++) {
    int mm = tid + r * nthreads;
    if (mm < 1) mm = 1;
    if (mm > (1 << LB) - 1) mm = (1 << LB) - 1;
    const int sl = 31 - __builtin_clz(mm);
    v[r] = tw[(((1 << S0) + b) << sl) + (mm - (1 << sl))];
  }
}

// K1 (multiply): c2_j = a1 b1, inverse transform in LDS, first radix-16 pass of the forward transforms modulo the other primes
// K1 (key switch, KS = true): the operand limb itself (NTT form, Galois gather folded into the load) instead of a1 b1
template <int LB, bool GUARD, bool KS, bool GAL>
__global__ __launch_bounds__((1 << LB) / 16) void k_isplit_pass0(DevCtx c, const u64 *__restrict__ a, const u64 *__restrict__ b,
                                                                 size_t a_stride, u64 *__restrict__ part, int nl, u32 gelt, u32 fpmask) {
  static_assert(LB == 14, "split transforms are laid out for N = 2^14");
  __shared__ u64 lds[lds_words(LB)];
  const int j = blockIdx.x % nl;
  const size_t ct = blockIdx.x / nl;
  const size_t N = (size_t)1 << LB, pw = (size_t)nl * N;
  const Mod m = c.mods[j];
  u64 src[16];
  if ((fpmask >> j) & 1u) {  // q_j below 2^50: this limb's own inverse transform in exact fp64 (workgroup-uniform)
    double *ldsd = reinterpret_cast<double *>(lds);
    const Mod mf = mod_at(c, j);
    const FpTable tf = fp_table(c, j);
    const double q = mf.qd, qinv = mf.qinv;
    auto canon = [&](int r, int, double v) { src[r] = fp_to_canon(fp_mul_lazy(v, mf.inv_n_c, mf.inv_n_cq, q), q, qinv); };
    if (KS) {
      const u64 *__restrict__ sp = a + ct * a_stride + (size_t)j * N;
      ntt_inv_block_a<LB, FpArith>(ldsd, [&](int, int i) { return fp_from_u64(sp[galois_ntt_src<GAL>((u32)i, gelt, LB)]); }, canon, tf, mf, 0, 0);
    } else {
      const u64 *__restrict__ a1 = a + ct * 2 * pw + pw + (size_t)j * N, *__restrict__ b1 = b + ct * 2 * pw + pw + (size_t)j * N;
      ntt_inv_block_a<LB, FpArith>(ldsd, [&](int, int i) { return i_fp_mulmod(fp_from_u64(a1[i]), fp_from_u64(b1[i]), q, qinv); }, canon, tf,
                                   mf, 0, 0);
    }
  } else {
    const NttTable t = ntt_table(c, j);
    if (KS) {
      const u64 *__restrict__ sp = a + ct * a_stride + (size_t)j * N;
      ntt_inv_block<LB>(
          lds, [&](int, int i) { return sp[galois_ntt_src<GAL>((u32)i, gelt, LB)]; },
          [&](int r, int, u64 v) { src[r] = scale_inv_n(v, m); }, t, m, 0, 0);
    } else {
      const u64 *__restrict__ a1 = a + ct * 2 * pw + pw + (size_t)j * N, *__restrict__ b1 = b + ct * 2 * pw + pw + (size_t)j * N;
      ntt_inv_block<LB>(
          lds, [&](int, int i) { return mul_mod(a1[i], b1[i], m); }, [&](int r, int, u64 v) { src[r] = scale_inv_n(v, m); }, t, m, 0,
          0);
    }
  }
  const int tid = threadIdx.x;
  const int hi0[1] = {0};
  using A = IntArith<GUARD>;
  for (int I = 0; I <= nl; I++) {
    if (I == j) continue;
    const int ki = (I == nl) ? c.K - 1 : I;
    const Mod mI = c.mods[ki];
    u64 *__restrict__ dst = part + ((ct * (nl + 1) + I) * nl + j) * (size_t)c.ps;
    if (I < nl && ((fpmask >> I) & 1u)) {  // q_I below 2^50: canonical residue modulo q_I as a double, fp64 register pass, raw doubles out
      const Mod mf = mod_at(c, ki);
      const FpTable tf = fp_table(c, ki);
      const FpK kf = FpArith::consts(mf);
      double yd[16];
#pragma unroll
      for (int k = 0; k < 16; k++) yd[k] = fp_from_u64(reduce64(src[k], mI));
      fwd_pass<FpArith, LB, 0, 4>(yd, hi0, tf, kf, 0, 0);
#pragma unroll
      for (int k = 0; k < 16; k++) reinterpret_cast<double *>(dst)[(k << 10) + tid] = yd[k];
      continue;
    }
    const NttTable t = ntt_table(c, ki);
    const typename A::K kk = A::consts(mI);
    // residues are < q_j; the guarded pass accepts inputs < 4 q_I, the unguarded one < 8 q_I (workgroup-uniform choice)
    const bool need_reduce = GUARD ? (m.q > mI.q) : ((m.q >> 3) >= mI.q);
    u64 y[16];
    if (need_reduce) {
#pragma unroll
      for (int k = 0; k < 16; k++) y[k] = reduce64(src[k], mI);
    } else {
#pragma unroll
      for (int k = 0; k < 16; k++) y[k] = src[k];
    }
    fwd_pass<A, LB, 0, 4>(y, hi0, t, kk, 0, 0);
#pragma unroll
    for (int k = 0; k < 16; k++) dst[(k << 10) + tid] = y[k];
  }
}

// ---- N = 2^15: a limb is 256 KiB, nothing is LDS-resident -- the first step in two kernels (cf. abc_kernels_gsplit.hip) ----
// K1a (limb j, block; four ciphertexts per workgroup, one wavefront each): operand block (a1 b1 for a multiply; the operand with
//     the Galois gather folded in for a key switch) -> stages LOGN-1..LB of the inverse transform -> hinv (values in [0, 2q))
template <int LOGN, int MODE, bool GAL>
__global__ __launch_bounds__(256) void k_igsplit_inv_tails(DevCtx c, const u64 *__restrict__ a, const u64 *__restrict__ b, size_t a_stride,
                                                           u64 *__restrict__ hinv, int nl, int cc, u32 gelt) {
  constexpr int LOGNB = LOGN - 10, NB = 1 << LOGNB;
  extern __shared__ u64 dynu[];  // 4 transform buffers, then the block's inverse-twiddle table
  const int W = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int blk = blockIdx.x & (NB - 1);
  const int j = (int)((blockIdx.x >> LOGNB) % (unsigned)nl);
  const size_t ct = (size_t)((blockIdx.x >> LOGNB) / (unsigned)nl) * 4 + (size_t)W;
  const size_t N = (size_t)1 << LOGN, base = (size_t)blk << 10, PS = (size_t)c.ps, pw = (size_t)nl * N;
  const Mod m = c.mods[j];
  const NttTable t = ntt_table(c, j);
  u64x2 *litw = reinterpret_cast<u64x2 *>(dynu + 4 * lds_words(10));
  u64x2 twv[4];
  block_twiddles_fetch_g<10, u64x2, 4>(t.itw, LOGNB, blk, (int)threadIdx.x, 256, twv);
  block_twiddles_store<10, u64x2, 4>(litw, (int)threadIdx.x, 256, twv);
  __syncthreads();
  if (ct >= (size_t)cc) return;  // wavefront-uniform, after the only workgroup barrier
  u64 *buf = dynu + W * lds_words(10);
  u64 *__restrict__ dst = hinv + (ct * nl + j) * PS + base;
  auto st = [&](int, int i, u64 v) { dst[i] = v; };
  if (MODE == 0) {
    const u64 *__restrict__ a1 = a + ct * 2 * pw + pw + (size_t)j * N + base, *__restrict__ b1 = b + ct * 2 * pw + pw + (size_t)j * N + base;
    auto ld = [&](int, int i) { return mul_mod(a1[i], b1[i], m); };
    ntt_inv_block_a<10, IntArith<true>, decltype(ld), decltype(st), true>(buf, ld, st, t, m, LOGNB, blk, lane, litw);
  } else {
    const u64 *__restrict__ sp = a + ct * a_stride + (size_t)j * N;
    auto ld = [&](int, int i) { return sp[galois_ntt_src<GAL>((u32)(base + i), gelt, LOGN)]; };
    ntt_inv_block_a<10, IntArith<true>, decltype(ld), decltype(st), true>(buf, ld, st, t, m, LOGNB, blk, lane, litw);
  }
}
// K1b (registers only): the cross pass of that inverse transform, N^-1, canonical coefficient; per other key prime the forward
//     cross pass -> half-done decomposition limbs `part`
template <int LOGN, bool GUARD>
__global__ __launch_bounds__(256) void k_igsplit_cross(DevCtx c, const u64 *__restrict__ hinv, u64 *__restrict__ part, int nl, u32 fpmask) {
  constexpr int LOGNB = LOGN - 10, NB = 1 << LOGNB;
  const int p = (int)((blockIdx.x & 3) << 8) + (int)threadIdx.x;
  const int j = (int)((blockIdx.x >> 2) % (unsigned)nl);
  const size_t ct = (size_t)((blockIdx.x >> 2) / (unsigned)nl);
  const size_t PS = (size_t)c.ps;
  const Mod m = c.mods[j];
  u64 x[NB];
  {
    const NttTable t = ntt_table(c, j);
    const IntArith<true>::K ks = IntArith<true>::consts(m);
    const u64 *__restrict__ src = hinv + (ct * nl + j) * PS;
#pragma unroll
    for (int k = 0; k < NB; k++) x[k] = src[(k << 10) + p];
    x_inv_cross<LOGNB, IntArith<true>>(x, t, ks);
#pragma unroll
    for (int k = 0; k < NB; k++) x[k] = scale_inv_n(x[k], m);  // canonical [0, q_j)
  }
  using A = IntArith<GUARD>;
  for (int I = 0; I <= nl; I++) {
    if (I == j) continue;
    const int ki = (I == nl) ? c.K - 1 : I;
    const Mod mI = c.mods[ki];
    u64 *__restrict__ dst = part + ((ct * (nl + 1) + I) * nl + j) * PS;
    if (I < nl && ((fpmask >> I) & 1u)) {  // q_I below 2^50: fp64 cross pass, raw doubles out (as k_isplit_pass0)
      const Mod mf = mod_at(c, ki);
      const FpTable tf = fp_table(c, ki);
      const FpK kf = FpArith::consts(mf);
      double yd[NB];
#pragma unroll
      for (int k = 0; k < NB; k++) yd[k] = fp_from_u64(reduce64(x[k], mI));
      x_fwd_cross<LOGNB, FpArith>(yd, tf, kf);
#pragma unroll
      for (int k = 0; k < NB; k++) reinterpret_cast<double *>(dst)[(k << 10) + p] = yd[k];
      continue;
    }
    const NttTable t = ntt_table(c, ki);
    const typename A::K kk = A::consts(mI);
    // unguarded: 4 q per stage over LOGN stages on top of the input must stay below 64 q -- input below 2 q_I at 15 stages
    const bool need_reduce = GUARD ? (m.q > mI.q) : ((m.q >> (LOGN == 14 ? 3 : 1)) >= mI.q);
    u64 y[NB];
    if (need_reduce) {
#pragma unroll
      for (int k = 0; k < NB; k++) y[k] = reduce64(x[k], mI);
    } else {
#pragma unroll
      for (int k = 0; k < NB; k++) y[k] = x[k];
    }
    x_fwd_cross<LOGNB, A>(y, t, kk);
#pragma unroll
    for (int k = 0; k < NB; k++) dst[(k << 10) + p] = y[k];
  }
}

// K2a: the special prime's inner product and the block-local stages of its inverse transform (cf. k_split_special_fp)
template <bool GUARD, int NL, int LOGN>
__global__ __launch_bounds__(NL * 64) void k_isplit_special(DevCtx c, const u64 *__restrict__ part, const u64 *__restrict__ key,
                                                            u64 *__restrict__ tsp_half) {
  extern __shared__ u64 dynu[];  // max(nl, 2) buffers of one 1024-point block
  constexpr int nl = NL, LOGNB = LOGN - 10, NB = 1 << LOGNB;
  using A = IntArith<GUARD>;
  const int J = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int blk = blockIdx.x & (NB - 1);
  const size_t ct = (size_t)(blockIdx.x >> LOGNB);
  const size_t N = (size_t)c.n, base = (size_t)blk << 10, PS = (size_t)c.ps;
  const int ki = c.K - 1;
  const Mod m = c.mods[ki];
  const NttTable t = ntt_table(c, ki);
  {
    u64 *buf = dynu + J * lds_words(10);
    const u64 *__restrict__ src = part + ((ct * (nl + 1) + nl) * nl + J) * PS + base;
    ntt_fwd_block_a<10, A>(
        buf, [&](int, int i) { return src[i]; }, [&](int, int i, u64 v) { buf[lds_pad(i)] = canon_fwd<GUARD>(v, m); }, t, m, LOGNB, blk,
        lane);
  }
  __syncthreads();
  for (int e = 2 * (int)threadIdx.x; e < 1024; e += 2 * (int)blockDim.x) {
    U128 a00{0, 0}, a01{0, 0}, a10{0, 0}, a11{0, 0};
#pragma unroll
    for (int Jx = 0; Jx < NL; Jx++) {
      const u64x2 x = *reinterpret_cast<const u64x2 *>(dynu + Jx * lds_words(10) + lds_pad(e));
      const u64x2 k0 = *reinterpret_cast<const u64x2 *>(key + (((size_t)Jx * 2 + 0) * c.K + ki) * N + base + e);
      const u64x2 k1 = *reinterpret_cast<const u64x2 *>(key + (((size_t)Jx * 2 + 1) * c.K + ki) * N + base + e);
      mac128(a00, x.x, k0.x); mac128(a01, x.y, k0.y);
      mac128(a10, x.x, k1.x); mac128(a11, x.y, k1.y);
      if ((Jx & 3) == 3 || Jx == NL - 1) {
        a00 = U128{barrett_reduce(a00, m), 0}; a01 = U128{barrett_reduce(a01, m), 0};
        a10 = U128{barrett_reduce(a10, m), 0}; a11 = U128{barrett_reduce(a11, m), 0};
      }
    }
    // park the two sums in buffers 0 and 1: a thread overwrites only the words it alone has read
    *reinterpret_cast<u64x2 *>(dynu + lds_pad(e)) = u64x2{a00.lo, a01.lo};
    *reinterpret_cast<u64x2 *>(dynu + lds_words(10) + lds_pad(e)) = u64x2{a10.lo, a11.lo};
  }
  __syncthreads();
  for (int comp = J; comp < 2; comp += nl) {  // stages 13..4 of the special-prime limb's inverse transform, one wavefront each
    u64 *buf = dynu + comp * lds_words(10);
    u64 *__restrict__ dst = tsp_half + (ct * 2 + comp) * PS + base;
    ntt_inv_block_a<10, IntArith<true>>(
        buf, [&](int, int i) { return buf[lds_pad(i)]; }, [&](int, int i, u64 v) { dst[i] = v; }, t, m, LOGNB, blk, lane);
  }
}

// K2b (registers only): last radix-16 pass of that inverse transform, N^-1, + q_sp/2; then per data prime the first radix-16
// pass of the forward transform of (t mod q_j + fix)
template <int LB, bool GUARD>
__global__ __launch_bounds__(256) void k_isplit_pass(DevCtx c, const u64 *__restrict__ tsp_half, u64 *__restrict__ tpart, int nl,
                                                     u32 fpmask) {
  static_assert(LB == 14 || LB == 15, "split transforms: 16 or 32 blocks of 1024 points");
  constexpr int LOGNB = LB - 10, NB = 1 << LOGNB;
  const size_t cc = blockIdx.x >> 2;
  const int p = (int)((blockIdx.x & 3) << 8) + (int)threadIdx.x;
  const size_t PS = (size_t)c.ps;
  u64 x[NB];
  const Mod ms = c.mods[c.K - 1];
  {
    const NttTable ts = ntt_table(c, c.K - 1);
    const IntArith<true>::K ks = IntArith<true>::consts(ms);
    const u64 *__restrict__ src = tsp_half + cc * PS;
#pragma unroll
    for (int k = 0; k < NB; k++) x[k] = src[(k << 10) + p];
    x_inv_cross<LOGNB, IntArith<true>>(x, ts, ks);
    const u64 half = ms.q >> 1;
#pragma unroll
    for (int k = 0; k < NB; k++) x[k] = add_mod(scale_inv_n(x[k], ms), half, ms.q);  // canonical [0, q_sp)
  }
  using A = IntArith<GUARD>;
  for (int j = 0; j < nl; j++) {
    const Mod m = c.mods[j];
    const u64 hm = reduce64(ms.q >> 1, m);
    const u64 fix = hm ? m.q - hm : 0;
    u64 *__restrict__ dst = tpart + (cc * nl + j) * PS;
    if ((fpmask >> j) & 1u) {
      const Mod mf = mod_at(c, j);
      const FpTable tf = fp_table(c, j);
      const FpK kf = FpArith::consts(mf);
      double yd[NB];
#pragma unroll
      for (int k = 0; k < NB; k++) yd[k] = fp_from_u64(add_mod(reduce64(x[k], m), fix, m.q));
      x_fwd_cross<LOGNB, FpArith>(yd, tf, kf);
#pragma unroll
      for (int k = 0; k < NB; k++) reinterpret_cast<double *>(dst)[(k << 10) + p] = yd[k];
      continue;
    }
    const NttTable t = ntt_table(c, j);
    const typename A::K kk = A::consts(m);
    u64 y[NB];
#pragma unroll
    for (int k = 0; k < NB; k++) y[k] = add_mod(reduce64(x[k], m), fix, m.q);
    x_fwd_cross<LOGNB, A>(y, t, kk);
#pragma unroll
    for (int k = 0; k < NB; k++) dst[(k << 10) + p] = y[k];
  }
}

// K2c (cf. k_split4_main_fp): decomposition tails + mod-down tails + inner product + (sum + q_sp c - NTT(t)) q_sp^-1
template <int MODE, int NL>
struct IPairOps {
  u64x2 k0[NL], k1[NL];
  u64x2 a0, a1, b0, b1;       // MODE 0
  u64 xs[2], d0s[2], d1s[2];  // MODE 1
};

template <int MODE, bool GAL, int NL, bool GUARD, int LOGN>
__global__ __launch_bounds__(512, NL <= 4 ? 4 : 2) void k_isplit_main(DevCtx c, const u64 *__restrict__ part, const u64 *__restrict__ tpart,
                                                        const u64 *__restrict__ opa, const u64 *__restrict__ opb, size_t opa_stride,
                                                        size_t opb_stride, int add_c1, const u64 *__restrict__ key, u64 *__restrict__ out,
                                                        u32 gelt, u32 imap, int ni) {
  // grid (ct, slot, block), slot < ni; the data prime of a slot is nibble `slot` of imap (all of them: 0x76543210, ni = nl)
  extern __shared__ u64 dynu[];  // nl + 1 transform buffers, then the block's twiddle table (1024 {w, Shoup} pairs)
  static_assert(NL + 1 <= 8, "one wavefront per limb, eight wavefronts");
  constexpr int nl = NL, NT = 512, PER = 2, LOGNB = LOGN - 10, NB = 1 << LOGNB;
  using A = IntArith<GUARD>;
  const int W = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int blk = blockIdx.x & (NB - 1);
  const int I = (int)((imap >> (4 * ((blockIdx.x >> LOGNB) % (unsigned)ni))) & 15u);
  const size_t ct = (size_t)((blockIdx.x >> LOGNB) / (unsigned)ni);
  const size_t N = (size_t)c.n, base = (size_t)blk << 10, PS = (size_t)c.ps;
  const Mod m = c.mods[I];
  const NttTable t = ntt_table(c, I);
  u64x2 *ltw = reinterpret_cast<u64x2 *>(dynu + (nl + 1) * lds_words(10));
  const size_t pw = (size_t)nl * N;
  const ABC_CONST_AS DevConst *cst = (const ABC_CONST_AS DevConst *)c.cst;
  const u64 sp = cst->special_mod_q[I], sp_s = cst->special_mod_q_s[I];
  const u64 inv = cst->inv_special[I], inv_s = cst->inv_special_s[I];

  u64x2 twv[PER];
  block_twiddles_fetch_g<10, u64x2, PER>(t.tw, LOGNB, blk, (int)threadIdx.x, NT, twv);
  const bool has_limb = W <= nl;
  const int Wc = has_limb ? W : 0;
  const u64 *__restrict__ src = (Wc < nl - 1) ? part + ((ct * (nl + 1) + I) * nl + (Wc < I ? Wc : Wc + 1)) * PS + base
                                              : tpart + ((ct * 2 + (Wc - (nl - 1))) * nl + I) * PS + base;
  u64 xin[16];
  if (has_limb) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const u64x2 v = *reinterpret_cast<const u64x2 *>(src + (k << 7) + 2 * lane);
      xin[k] = v.x;
      xin[8 + k] = v.y;
    }
  }
  const int e = 2 * (int)threadIdx.x;
  IPairOps<MODE, NL> o;
#pragma unroll
  for (int Jx = 0; Jx < NL; Jx++) {
    o.k0[Jx] = *reinterpret_cast<const u64x2 *>(key + (((size_t)Jx * 2 + 0) * c.K + I) * N + base + e);
    o.k1[Jx] = *reinterpret_cast<const u64x2 *>(key + (((size_t)Jx * 2 + 1) * c.K + I) * N + base + e);
  }
  if (MODE == 0) {
    const u64 *pa = opa + ct * 2 * pw + (size_t)I * N + base + e, *pb = opb + ct * 2 * pw + (size_t)I * N + base + e;
    o.a0 = *reinterpret_cast<const u64x2 *>(pa); o.a1 = *reinterpret_cast<const u64x2 *>(pa + pw);
    o.b0 = *reinterpret_cast<const u64x2 *>(pb); o.b1 = *reinterpret_cast<const u64x2 *>(pb + pw);
  } else {
    const u64 *xl = opa + ct * opa_stride + (size_t)I * N;
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const u32 si = galois_ntt_src<GAL>((u32)(base + e + k), gelt, c.logn);
      o.xs[k] = xl[si];
      o.d0s[k] = o.d1s[k] = 0;
      if (opb) {
        const u64 *ad = opb + ct * opb_stride + (size_t)I * N;
        o.d0s[k] = ad[si];
        if (add_c1) o.d1s[k] = ad[pw + si];
      }
    }
  }
  block_twiddles_store<10, u64x2, PER>(ltw, (int)threadIdx.x, NT, twv);
  __syncthreads();
  if (has_limb) {
    u64 *buf = dynu + W * lds_words(10);
    ntt_fwd_tail1024_pairs<A>(buf, xin, [&](int, int i, u64 v) { buf[lds_pad(i)] = canon_fwd<GUARD>(v, m); }, t, m, LOGNB, blk, lane, ltw);
  }
  __syncthreads();
  const u64 *tt0 = dynu + (nl - 1) * lds_words(10), *tt1 = dynu + nl * lds_words(10);
  U128 s0[2] = {{0, 0}, {0, 0}}, s1[2] = {{0, 0}, {0, 0}};
  u64 d0[2] = {0, 0}, d1[2] = {0, 0};
#pragma unroll
  for (int Jx = 0; Jx < NL; Jx++) {
    u64 x[2];
    if (Jx == I) {
      if (MODE == 0) {
        const u64 x0[2] = {o.a0.x, o.a0.y}, x1[2] = {o.a1.x, o.a1.y}, y0[2] = {o.b0.x, o.b0.y}, y1[2] = {o.b1.x, o.b1.y};
#pragma unroll
        for (int k = 0; k < 2; k++) {
          x[k] = mul_mod(x1[k], y1[k], m);
          d0[k] = mul_mod(x0[k], y0[k], m);
          U128 acc = mul_wide(x0[k], y1[k]);
          mac128(acc, x1[k], y0[k]);
          d1[k] = barrett_reduce(acc, m);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 2; k++) {
          x[k] = o.xs[k];
          d0[k] = o.d0s[k];
          d1[k] = o.d1s[k];
        }
      }
    } else {
      const int w = Jx < I ? Jx : Jx - 1;
      const u64x2 v = *reinterpret_cast<const u64x2 *>(dynu + w * lds_words(10) + lds_pad(e));
      x[0] = v.x;
      x[1] = v.y;
    }
    mac128(s0[0], x[0], o.k0[Jx].x); mac128(s0[1], x[1], o.k0[Jx].y);
    mac128(s1[0], x[0], o.k1[Jx].x); mac128(s1[1], x[1], o.k1[Jx].y);
    if ((Jx & 3) == 3 && Jx != NL - 1) {
#pragma unroll
      for (int k = 0; k < 2; k++) {
        s0[k] = U128{barrett_reduce(s0[k], m), 0};
        s1[k] = U128{barrett_reduce(s1[k], m), 0};
      }
    }
  }
  const u64x2 u0 = *reinterpret_cast<const u64x2 *>(tt0 + lds_pad(e)), u1 = *reinterpret_cast<const u64x2 *>(tt1 + lds_pad(e));
  const u64 t0[2] = {u0.x, u0.y}, t1[2] = {u1.x, u1.y};
  u64 r0[2], r1[2];
#pragma unroll
  for (int k = 0; k < 2; k++) {
    // (sum + q_sp (c0, c1) - NTT(t)) q_sp^-1
    const u64 a0v = add_mod(barrett_reduce(s0[k], m), mul_shoup(d0[k], sp, sp_s, m.q), m.q);
    const u64 a1v = add_mod(barrett_reduce(s1[k], m), mul_shoup(d1[k], sp, sp_s, m.q), m.q);
    r0[k] = mul_shoup(sub_mod(a0v, t0[k], m.q), inv, inv_s, m.q);
    r1[k] = mul_shoup(sub_mod(a1v, t1[k], m.q), inv, inv_s, m.q);
  }
  *reinterpret_cast<u64x2 *>(out + ((ct * 2 + 0) * nl + I) * N + base + e) = u64x2{r0[0], r0[1]};
  *reinterpret_cast<u64x2 *>(out + ((ct * 2 + 1) * nl + I) * N + base + e) = u64x2{r1[0], r1[1]};
}

// K2c for deep chains (8 to 15 data limbs; cf. k_gsplit_main_deep): sixteen wavefronts, one per limb, key words loaded inside the sum
template <int MODE, bool GAL, bool GUARD, int LOGN>
__global__ __launch_bounds__(1024) void k_isplit_main_deep(DevCtx c, const u64 *__restrict__ part, const u64 *__restrict__ tpart,
                                                           const u64 *__restrict__ opa, const u64 *__restrict__ opb, size_t opa_stride,
                                                           size_t opb_stride, int add_c1, const u64 *__restrict__ key, u64 *__restrict__ out,
                                                           u32 gelt, int nl, u64 imap, int ni) {
  extern __shared__ u64 dynu[];  // nl + 1 transform buffers, then the block's twiddle table
  constexpr int NT = 1024, LOGNB = LOGN - 10, NB = 1 << LOGNB;
  using A = IntArith<GUARD>;
  const int W = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int blk = blockIdx.x & (NB - 1);
  const int I = (int)((imap >> (4 * ((blockIdx.x >> LOGNB) % (unsigned)ni))) & 15u);
  const size_t ct = (size_t)((blockIdx.x >> LOGNB) / (unsigned)ni);
  const size_t N = (size_t)c.n, base = (size_t)blk << 10, PS = (size_t)c.ps;
  const Mod m = c.mods[I];
  const NttTable t = ntt_table(c, I);
  u64x2 *ltw = reinterpret_cast<u64x2 *>(dynu + (nl + 1) * lds_words(10));
  const size_t pw = (size_t)nl * N;
  const ABC_CONST_AS DevConst *cst = (const ABC_CONST_AS DevConst *)c.cst;
  const u64 sp = cst->special_mod_q[I], sp_s = cst->special_mod_q_s[I];
  const u64 inv = cst->inv_special[I], inv_s = cst->inv_special_s[I];
  u64x2 twv[1];
  block_twiddles_fetch_g<10, u64x2, 1>(t.tw, LOGNB, blk, (int)threadIdx.x, NT, twv);
  const bool has_limb = W <= nl;
  const int Wc = has_limb ? W : 0;
  const u64 *__restrict__ src = (Wc < nl - 1) ? part + ((ct * (nl + 1) + I) * nl + (Wc < I ? Wc : Wc + 1)) * PS + base
                                              : tpart + ((ct * 2 + (Wc - (nl - 1))) * nl + I) * PS + base;
  u64 xin[16];
  if (has_limb) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const u64x2 v = *reinterpret_cast<const u64x2 *>(src + (k << 7) + 2 * lane);
      xin[k] = v.x;
      xin[8 + k] = v.y;
    }
  }
  block_twiddles_store<10, u64x2, 1>(ltw, (int)threadIdx.x, NT, twv);
  __syncthreads();
  if (has_limb) {
    u64 *buf = dynu + W * lds_words(10);
    ntt_fwd_tail1024_pairs<A>(buf, xin, [&](int, int i, u64 v) { buf[lds_pad(i)] = canon_fwd<GUARD>(v, m); }, t, m, LOGNB, blk, lane, ltw);
  }
  __syncthreads();
  if (threadIdx.x >= 512) return;
  const int e = 2 * (int)threadIdx.x;
  const u64 *tt0 = dynu + (nl - 1) * lds_words(10), *tt1 = dynu + nl * lds_words(10);
  U128 s0[2] = {{0, 0}, {0, 0}}, s1[2] = {{0, 0}, {0, 0}};
  u64 d0[2] = {0, 0}, d1[2] = {0, 0};
  for (int Jx = 0; Jx < nl; Jx++) {
    u64 x[2];
    if (Jx == I) {
      if (MODE == 0) {
        const u64 *pa = opa + ct * 2 * pw + (size_t)I * N + base + e, *pb = opb + ct * 2 * pw + (size_t)I * N + base + e;
        const u64x2 a0 = *reinterpret_cast<const u64x2 *>(pa), a1 = *reinterpret_cast<const u64x2 *>(pa + pw);
        const u64x2 b0 = *reinterpret_cast<const u64x2 *>(pb), b1 = *reinterpret_cast<const u64x2 *>(pb + pw);
        const u64 x0[2] = {a0.x, a0.y}, x1[2] = {a1.x, a1.y}, y0[2] = {b0.x, b0.y}, y1[2] = {b1.x, b1.y};
#pragma unroll
        for (int k = 0; k < 2; k++) {
          x[k] = mul_mod(x1[k], y1[k], m);
          d0[k] = mul_mod(x0[k], y0[k], m);
          U128 acc = mul_wide(x0[k], y1[k]);
          mac128(acc, x1[k], y0[k]);
          d1[k] = barrett_reduce(acc, m);
        }
      } else {
        const u64 *xl = opa + ct * opa_stride + (size_t)I * N;
#pragma unroll
        for (int k = 0; k < 2; k++) {
          const u32 si = galois_ntt_src<GAL>((u32)(base + e + k), gelt, c.logn);
          x[k] = xl[si];
          if (opb) {
            const u64 *ad = opb + ct * opb_stride + (size_t)I * N;
            d0[k] = ad[si];
            if (add_c1) d1[k] = ad[pw + si];
          }
        }
      }
    } else {
      const int w = Jx < I ? Jx : Jx - 1;
      const u64x2 v = *reinterpret_cast<const u64x2 *>(dynu + w * lds_words(10) + lds_pad(e));
      x[0] = v.x;
      x[1] = v.y;
    }
    const u64x2 k0 = *reinterpret_cast<const u64x2 *>(key + (((size_t)Jx * 2 + 0) * c.K + I) * N + base + e);
    const u64x2 k1 = *reinterpret_cast<const u64x2 *>(key + (((size_t)Jx * 2 + 1) * c.K + I) * N + base + e);
    mac128(s0[0], x[0], k0.x); mac128(s0[1], x[1], k0.y);
    mac128(s1[0], x[0], k1.x); mac128(s1[1], x[1], k1.y);
    if ((Jx & 3) == 3) {  // four products of canonical operands per 128-bit accumulator
#pragma unroll
      for (int k = 0; k < 2; k++) {
        s0[k] = U128{barrett_reduce(s0[k], m), 0};
        s1[k] = U128{barrett_reduce(s1[k], m), 0};
      }
    }
  }
  const u64x2 u0 = *reinterpret_cast<const u64x2 *>(tt0 + lds_pad(e)), u1 = *reinterpret_cast<const u64x2 *>(tt1 + lds_pad(e));
  const u64 t0[2] = {u0.x, u0.y}, t1[2] = {u1.x, u1.y};
  u64 r0[2], r1[2];
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const u64 a0v = add_mod(barrett_reduce(s0[k], m), mul_shoup(d0[k], sp, sp_s, m.q), m.q);
    const u64 a1v = add_mod(barrett_reduce(s1[k], m), mul_shoup(d1[k], sp, sp_s, m.q), m.q);
    r0[k] = mul_shoup(sub_mod(a0v, t0[k], m.q), inv, inv_s, m.q);
    r1[k] = mul_shoup(sub_mod(a1v, t1[k], m.q), inv, inv_s, m.q);
  }
  *reinterpret_cast<u64x2 *>(out + ((ct * 2 + 0) * nl + I) * N + base + e) = u64x2{r0[0], r0[1]};
  *reinterpret_cast<u64x2 *>(out + ((ct * 2 + 1) * nl + I) * N + base + e) = u64x2{r1[0], r1[1]};
}

// steps 2-4 of a deep chain at N = 2^15: special prime (integer), register pass, then the last step per arithmetic class
template <bool GUARD>
static void launch_isplit_tail_deep15(hipStream_t st, abc_hip_ctx *c, size_t cc, int nl, u64 *part, u64 *tpart, u64 *tsp_half, int mode,
                                      const u64 *opa, const u64 *opb, size_t opa_stride, size_t opb_stride, int add_c1, const u64 *key,
                                      u64 *out, u32 gelt, u32 fpmask) {
  const size_t lds_sp = (size_t)(nl * lds_words(10)) * 8;
  const size_t lds_main = (size_t)((nl + 1) * lds_words(10)) * 8 + 1024 * 16;
  u64 imap_int = 0, imap_fp = 0;
  int ni_int = 0, ni_fp = 0;
  for (int I = 0; I < nl; I++) {
    if ((fpmask >> I) & 1u) imap_fp |= (u64)I << (4 * ni_fp++);
    else imap_int |= (u64)I << (4 * ni_int++);
  }
  const dim3 gsp((unsigned)(cc * 32));
#define ABC_ISPD(NLV) hipLaunchKernelGGL((k_isplit_special<GUARD, NLV, 15>), gsp, dim3(64 * NLV), lds_sp, st, c->dc, part, key, tsp_half)
  switch (nl) {
    case 8: ABC_ISPD(8); break;
    case 9: ABC_ISPD(9); break;
    case 10: ABC_ISPD(10); break;
    case 11: ABC_ISPD(11); break;
    case 12: ABC_ISPD(12); break;
    case 13: ABC_ISPD(13); break;
    case 14: ABC_ISPD(14); break;
    default: ABC_ISPD(15); break;
  }
#undef ABC_ISPD
  hipLaunchKernelGGL((k_isplit_pass<15, GUARD>), dim3((unsigned)(cc * 2 * 4)), dim3(256), 0, st, c->dc, tsp_half, tpart, nl, fpmask);
  if (ni_int) {
    const dim3 gmain((unsigned)(cc * ni_int * 32));
    if (mode == 0)
      hipLaunchKernelGGL((k_isplit_main_deep<0, false, GUARD, 15>), gmain, dim3(1024), lds_main, st, c->dc, part, tpart, opa, opb, opa_stride,
                         opb_stride, add_c1, key, out, gelt, nl, imap_int, ni_int);
    else if (gelt)
      hipLaunchKernelGGL((k_isplit_main_deep<1, true, GUARD, 15>), gmain, dim3(1024), lds_main, st, c->dc, part, tpart, opa, opb, opa_stride,
                         opb_stride, add_c1, key, out, gelt, nl, imap_int, ni_int);
    else
      hipLaunchKernelGGL((k_isplit_main_deep<1, false, GUARD, 15>), gmain, dim3(1024), lds_main, st, c->dc, part, tpart, opa, opb, opa_stride,
                         opb_stride, add_c1, key, out, gelt, nl, imap_int, ni_int);
  }
  if (ni_fp)
    gsplit_main_deep_subset15(st, c, cc, nl, mode, (const double *)part, (const double *)tpart, opa, opb, opa_stride, opb_stride, add_c1, key,
                              out, gelt, imap_fp, ni_fp);
}

// ---- launch sequence on one chunk ----
template <bool GUARD, int LOGN>
static void launch_isplit_tail(hipStream_t st, abc_hip_ctx *c, size_t cc, int nl, u64 *part, u64 *tpart, u64 *tsp_half, int mode,
                               const u64 *opa, const u64 *opb, size_t opa_stride, size_t opb_stride, int add_c1, const u64 *key, u64 *out,
                               u32 gelt, u32 fpmask) {
  const size_t lds_sp = (size_t)((nl < 2 ? 2 : nl) * lds_words(10)) * 8;
  const size_t lds_main = (size_t)((nl + 1) * lds_words(10)) * 8 + 1024 * 16;
  // data primes of the integer main kernel / of the fp64 main kernel, one nibble each
  u32 imap_int = 0, imap_fp = 0;
  int ni_int = 0, ni_fp = 0;
  for (int I = 0; I < nl; I++) {
    if ((fpmask >> I) & 1u) imap_fp |= (u32)I << (4 * ni_fp++);
    else imap_int |= (u32)I << (4 * ni_int++);
  }
  constexpr int NB = 1 << (LOGN - 10);
  const dim3 gsp((unsigned)(cc * NB)), gmain((unsigned)(cc * ni_int * NB));
#define ABC_ISP(NLV)                                                                                                                  \
  hipLaunchKernelGGL((k_isplit_special<GUARD, NLV, LOGN>), gsp, dim3(64 * NLV), lds_sp, st, c->dc, part, key, tsp_half);              \
  hipLaunchKernelGGL((k_isplit_pass<LOGN, GUARD>), dim3((unsigned)(cc * 2 * 4)), dim3(256), 0, st, c->dc, tsp_half, tpart, nl, fpmask); \
  if (ni_int == 0) {                                                                                                                  \
  } else if (mode == 0)                                                                                                               \
    hipLaunchKernelGGL((k_isplit_main<0, false, NLV, GUARD, LOGN>), gmain, dim3(512), lds_main, st, c->dc, part, tpart, opa, opb, opa_stride, \
                       opb_stride, add_c1, key, out, gelt, imap_int, ni_int);                                                         \
  else if (gelt)                                                                                                                      \
    hipLaunchKernelGGL((k_isplit_main<1, true, NLV, GUARD, LOGN>), gmain, dim3(512), lds_main, st, c->dc, part, tpart, opa, opb, opa_stride, \
                       opb_stride, add_c1, key, out, gelt, imap_int, ni_int);                                                         \
  else                                                                                                                                \
    hipLaunchKernelGGL((k_isplit_main<1, false, NLV, GUARD, LOGN>), gmain, dim3(512), lds_main, st, c->dc, part, tpart, opa, opb, opa_stride, \
                       opb_stride, add_c1, key, out, gelt, imap_int, ni_int)
  switch (nl) {
    case 1: ABC_ISP(1); break;
    case 2: ABC_ISP(2); break;
    case 3: ABC_ISP(3); break;
    case 4: ABC_ISP(4); break;
    case 5: ABC_ISP(5); break;
    case 6: ABC_ISP(6); break;
    default: ABC_ISP(7); break;
  }
#undef ABC_ISP
  if (ni_fp && LOGN == 14)
    split4_main_subset(st, c, cc, nl, mode, (const double *)part, (const double *)tpart, opa, opb, opa_stride, opb_stride, add_c1, key, out,
                       gelt, imap_fp, ni_fp);
  else if (ni_fp)
    gsplit_main_subset15(st, c, cc, nl, mode, (const double *)part, (const double *)tpart, opa, opb, opa_stride, opb_stride, add_c1, key, out,
                         gelt, imap_fp, ni_fp);
}

// scratch (words, limb stride c->dc.ps): part nl(nl+1) | tpart 2 nl | tsp_half 2
// (N = 2^15: + hinv nl, the operand after the block stages of its inverse transform)
size_t isplit_scratch_words(const abc_hip_ctx *c, int nl) {
  return ((size_t)nl * (nl + 1) + 2 * (size_t)nl + 2 + (c->logn == 15 ? (size_t)nl : 0)) * (size_t)c->dc.ps;
}

bool isplit_applies(const abc_hip_ctx *c, int nl) {
  if ((c->logn != 14 && c->logn != 15) || c->scheme != 2 || c->sw.no_fused || c->sw.no_split || c->sw.no_isplit || nl < 1 ||
      nl > (c->logn == 15 ? 15 : 7))
    return false;
  if (c->logn == 15 && c->sw.no_gsplit) return false;  // one switch turns both split sequences of that ring off (A/B, tests)
  for (int j = 0; j < c->K; j++)
    if (c->h_mods[j].bits > 60) return false;
  return true;
}

// one chunk of a multiply (mode 0: opa = a, opb = b) or of a key switch (mode 1: opa = operand in NTT form, opb = addend)
int isplit_chunk(abc_hip_ctx *c, hipStream_t st, u64 *scratch, size_t cc, int nl, int mode, const u64 *opa, const u64 *opb,
                 size_t opa_stride, size_t opb_stride, int add_c1, const u64 *key, u64 *out, u32 gelt) {
  const size_t PS = (size_t)c->dc.ps;
  u64 *part = scratch, *tpart = part + cc * (size_t)nl * (nl + 1) * PS, *tsp = tpart + cc * 2 * (size_t)nl * PS;
  bool guard = false;
  for (int j = 0; j < c->K; j++) guard = guard || !unguarded_ok(c->h_mods[j].bits);
  // data primes below 2^50 take the fp64 kernels (ABC_HIP_NO_FP64 / ABC_HIP_NO_MIXED: integers throughout)
  u32 fpmask = 0;
  if (c->use_fp && !c->sw.no_mixed)
    for (int j = 0; j < nl; j++)
      if (fp_ok(c->h_mods[j].bits)) fpmask |= 1u << j;
  if (c->logn == 15) {
    u64 *hinv = tsp + cc * 2 * PS;
    const dim3 ga((unsigned)(((cc + 3) / 4) * nl * 32)), gb((unsigned)(cc * nl * 4));
    const size_t lds = (size_t)(4 * lds_words(10)) * 8 + 1024 * 16;
    if (mode == 0)
      hipLaunchKernelGGL((k_igsplit_inv_tails<15, 0, false>), ga, dim3(256), lds, st, c->dc, opa, opb, 0, hinv, nl, (int)cc, 0u);
    else if (gelt)
      hipLaunchKernelGGL((k_igsplit_inv_tails<15, 1, true>), ga, dim3(256), lds, st, c->dc, opa, nullptr, opa_stride, hinv, nl, (int)cc, gelt);
    else
      hipLaunchKernelGGL((k_igsplit_inv_tails<15, 1, false>), ga, dim3(256), lds, st, c->dc, opa, nullptr, opa_stride, hinv, nl, (int)cc, 0u);
    if (nl > 7) {
      if (guard) {
        hipLaunchKernelGGL((k_igsplit_cross<15, true>), gb, dim3(256), 0, st, c->dc, hinv, part, nl, fpmask);
        launch_isplit_tail_deep15<true>(st, c, cc, nl, part, tpart, tsp, mode, opa, opb, opa_stride, opb_stride, add_c1, key, out, gelt, fpmask);
      } else {
        hipLaunchKernelGGL((k_igsplit_cross<15, false>), gb, dim3(256), 0, st, c->dc, hinv, part, nl, fpmask);
        launch_isplit_tail_deep15<false>(st, c, cc, nl, part, tpart, tsp, mode, opa, opb, opa_stride, opb_stride, add_c1, key, out, gelt, fpmask);
      }
      ABC_HIP_CHECK(hipGetLastError());
      return 0;
    }
    if (guard) {
      hipLaunchKernelGGL((k_igsplit_cross<15, true>), gb, dim3(256), 0, st, c->dc, hinv, part, nl, fpmask);
      launch_isplit_tail<true, 15>(st, c, cc, nl, part, tpart, tsp, mode, opa, opb, opa_stride, opb_stride, add_c1, key, out, gelt, fpmask);
    } else {
      hipLaunchKernelGGL((k_igsplit_cross<15, false>), gb, dim3(256), 0, st, c->dc, hinv, part, nl, fpmask);
      launch_isplit_tail<false, 15>(st, c, cc, nl, part, tpart, tsp, mode, opa, opb, opa_stride, opb_stride, add_c1, key, out, gelt, fpmask);
    }
    ABC_HIP_CHECK(hipGetLastError());
    return 0;
  }
  const dim3 g1((unsigned)(cc * nl)), b1((1 << 14) / 16);
  if (guard) {
    if (mode == 0) hipLaunchKernelGGL((k_isplit_pass0<14, true, false, false>), g1, b1, 0, st, c->dc, opa, opb, 0, part, nl, 0u, fpmask);
    else if (gelt) hipLaunchKernelGGL((k_isplit_pass0<14, true, true, true>), g1, b1, 0, st, c->dc, opa, nullptr, opa_stride, part, nl, gelt, fpmask);
    else hipLaunchKernelGGL((k_isplit_pass0<14, true, true, false>), g1, b1, 0, st, c->dc, opa, nullptr, opa_stride, part, nl, 0u, fpmask);
    launch_isplit_tail<true, 14>(st, c, cc, nl, part, tpart, tsp, mode, opa, opb, opa_stride, opb_stride, add_c1, key, out, gelt, fpmask);
  } else {
    if (mode == 0) hipLaunchKernelGGL((k_isplit_pass0<14, false, false, false>), g1, b1, 0, st, c->dc, opa, opb, 0, part, nl, 0u, fpmask);
    else if (gelt) hipLaunchKernelGGL((k_isplit_pass0<14, false, true, true>), g1, b1, 0, st, c->dc, opa, nullptr, opa_stride, part, nl, gelt, fpmask);
    else hipLaunchKernelGGL((k_isplit_pass0<14, false, true, false>), g1, b1, 0, st, c->dc, opa, nullptr, opa_stride, part, nl, 0u, fpmask);
    launch_isplit_tail<false, 14>(st, c, cc, nl, part, tpart, tsp, mode, opa, opb, opa_stride, opb_stride, add_c1, key, out, gelt, fpmask);
  }
  ABC_HIP_CHECK(hipGetLastError());
  return 0;
}


}  // namespace abc
