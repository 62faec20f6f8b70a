// abc_kernels_bfv.hip -- BFV-specific kernels: BEHZ RNS multiplication, plaintext operations,
// BatchEncoder on the device, BFV decryption rounding.
//
// Reference call sites replaced:
//   src/runtime/SealCiphertext.cpp:104,:122        Evaluator::multiply(_inplace)   -> bfv_multiply
//   src/runtime/SealCiphertext.cpp:159,:196        Evaluator::multiply_plain       -> bfv_multiply_plain
//   src/runtime/SealCiphertext.cpp:134,:145,:175,:184  add_plain / sub_plain       -> bfv_addsub_plain
//   src/runtime/SealCiphertextFactory.cpp:130      BatchEncoder::encode            -> batch_encode
//   src/runtime/SealCiphertextFactory.cpp:151      BatchEncoder::decode            -> batch_decode
//   src/runtime/SealCiphertextFactory.cpp:150      Decryptor::decrypt (rounding)   -> k_bfv_decrypt_round
// BEHZ (Bajard-Eynard-Hasan-Zucca) full-RNS multiply as SEAL implements it: extend q -> Bsk u {m~}
// with a fast base conversion, Montgomery-reduce the q-overflow, tensor in both bases, scale by t,
// fast floor by q into Bsk, Shenoy-Kumaresan conversion back to q.  These are element-wise over
// coefficients (one lane = one coefficient, all limbs in registers), so they are HBM-streaming kernels.
#include <cstdlib>

#include "abc_context.hpp"

namespace abc {

static inline unsigned grid_for(size_t items, int block) {
  size_t g = (items + block - 1) / block;
  const size_t cap = 256 * 8 * 4;
  return (unsigned)(g < cap ? (g ? g : 1) : cap);
}

// sum_i a[i]*b[i] mod m with a[i] < 2^61, b[i] < m.q: flush the 128-bit accumulator every 4 terms
// (barrett_reduce precondition x < 2^(k+63)).
template <class FA, class FB>
__device__ __forceinline__ u64 dot_mod(int n, FA a, FB b, const Mod &m) {
  u64 res = 0;
  for (int i0 = 0; i0 < n; i0 += 4) {
    U128 acc{0, 0};
    const int i1 = (i0 + 4 < n) ? i0 + 4 : n;
    for (int i = i0; i < i1; i++) mac128(acc, a(i), b(i));
    res = add_mod(res, barrett_reduce(acc, m), m.q);
  }
  return res;
}

// Same sum with precomputed Shoup quotients of the constants: every term is one lazy Shoup product in [0, 2q) (any
// 64-bit operand), summed in 64 bits -- no 128-bit accumulator, no Barrett per group.  The running sum is brought
// back below 2q after every fourth term where 8q could pass 2^64 (q of 59..61 bits: the BEHZ auxiliary primes);
// smaller moduli take up to 32 terms unreduced.  Returns a lazy value: < 2q if FLUSH else < 2nq (callers feed it to
// mul_shoup, which accepts any 64-bit operand, or reduce it with reduce64 / canon_lazy2).
// NN > 0: the term count as a compile-time constant (NN == n): the loop then unrolls before the callers' register arrays are
// lowered -- with a run-time bound the arrays behind the lambdas were indexed dynamically and lived in scratch memory
// (k_behz_floor<8, 8>: 100 bytes per lane).
template <int NN = 0, class FA, class FW, class FS>
__device__ __forceinline__ u64 dot_shoup_lazy(int n_rt, FA a, FW w, FS ws, const Mod &m) {
  const bool flush = m.bits > 58;
  const int n = NN ? NN : n_rt;
  u64 acc = 0;
#pragma unroll
  for (int i = 0; i < n; i++) {
    acc += mul_shoup_lazy(a(i), w(i), ws(i), m.q);
    if (flush && (i & 3) == 3) acc = csub(csub(acc, m.two_q << 1), m.two_q);  // < 8q -> < 2q
  }
  if (flush) acc = csub(csub(acc, m.two_q << 1), m.two_q);
  return acc;
}
// lazy value of dot_shoup_lazy -> [0, q)
__device__ __forceinline__ u64 canon_dot(u64 v, const Mod &m) { return m.bits > 58 ? csub(v, m.q) : reduce64(v, m); }

// Constant rows of the conversion matrices, fetched only once the previous outer iteration's result exists.  The fully
// unrolled BEHZ kernels are one basic block; left alone, the compiler hoists EVERY iteration's scalar constant loads to
// the top, 104 SGPRs cannot hold an 8 x 9 matrix with its Shoup quotients, and the spill code (v_writelane / v_readlane)
// was 45 % of the floor kernel's instructions.  The empty asm makes the row pointer depend on `dep` (a VGPR result of
// the previous iteration), so each iteration's loads stay in their iteration.
template <class T>
__device__ __forceinline__ const ABC_CONST_AS T *row_after(const ABC_CONST_AS T *p, u64 dep) {
  asm volatile("" : "+s"(p) : "v"(dep));
  return p;
}

// ---- BEHZ steps (1)-(2): q -> Bsk with Montgomery reduction of the q-overflow ----
// in: [polys][L][N] coefficient form; out: [polys][nBsk][N]
// LT > 0: limb counts known at compile time (L = LT, nBsk = NBT + 1), so every loop unrolls and the conversion
// constants arrive in batched scalar loads instead of one dependent s_load per inner iteration (the runtime-bound
// form is latency-bound on exactly that: 0.94 ms -> see DESIGN.md).  LT = 0: any shape.
template <int LT, int NBT>
__global__ __launch_bounds__(256) void k_behz_extend(DevCtx c, const u64 *in, const u64 *in2, u64 *out, size_t polys) {
  const ABC_CONST_AS DevConst &k = *(const ABC_CONST_AS DevConst *)c.cst;  // constants never change: scalar loads
  const int L = LT ? LT : k.nq, nBsk = LT ? NBT + 1 : k.nBsk;
  const size_t items = polys * c.n;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const size_t p = it >> c.logn, x = it & (c.n - 1);
    // polynomials [0, polys/2) come from `in`, the rest from `in2` (both operands of a multiply in one launch)
    const bool second = in2 && p >= polys / 2;
    const u64 *src = second ? in2 : in;
    const size_t pp = second ? p - polys / 2 : p;
    u64 tmp[LT ? LT : kMaxLimbs];
    u32 mt = 0;
#pragma unroll
    for (int i = 0; i < L; i++) {
      // * m~ * (q/q_i)^-1 as one constant (canonical result, so identical to the two-step product)
      const u64 v = mul_shoup(src[(pp * L + i) * c.n + x], k.ext_q[i], k.ext_q_s[i], mod_at(c, i).q);
      tmp[i] = v;
      mt += (u32)v * (u32)k.q_to_mtilde[i];  // arithmetic mod 2^32
    }
    const u32 r32 = mt * (u32)k.neg_inv_q_mod_mtilde;
    u64 dep = tmp[L - 1];
#pragma unroll
    for (int j = 0; j < nBsk; j++) {
      const Mod m = mod_at(c, c.id_bsk + j);
      const ABC_CONST_AS u64 *row = row_after(&k.q_to_bsk[j][0], dep), *row_s = row_after(&k.q_to_bsk_s[j][0], dep);
      const u64 conv = dot_shoup_lazy<LT>(L, [&](int i) { return tmp[i]; }, [&](int i) { return row[i]; },
                                      [&](int i) { return row_s[i]; }, m);  // < 2p
      u64 r = r32;
      if (r32 >= 0x80000000u) r += m.q - 0x100000000ull;  // centred representative of r mod m~
      const u64 v = conv + mul_shoup_lazy(r, k.q_mod_bsk[j], k.q_mod_bsk_s[j], m.q);  // < 4p < 2^64
      dep = mul_shoup(v, k.inv_mtilde_mod_bsk[j], k.inv_mtilde_mod_bsk_s[j], m.q);
      out[(p * nBsk + j) * c.n + x] = dep;
    }
  }
}

// ---- dyadic tensor over an arbitrary limb set: a,b [count][2][nlm][N] -> d [count][3][nlm][N] ----
__global__ __launch_bounds__(256) void k_tensor_map(DevCtx c, const u64 *a, const u64 *b, u64 *d, LimbMap map, int nlm,
                                                    size_t count) {
  const size_t pw = (size_t)nlm * c.n;
  const size_t items = count * pw;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const size_t ct = it / pw, w = it % pw;
    const Mod m = c.mods[map.id[w >> c.logn]];
    const u64 *pa = a + ct * 2 * pw + w, *pb = b + ct * 2 * pw + w;
    const u64 a0 = pa[0], a1 = pa[pw], b0 = pb[0], b1 = pb[pw];
    u64 *po = d + ct * 3 * pw + w;
    po[0] = mul_mod(a0, b0, m);
    po[pw] = add_mod(mul_mod(a0, b1, m), mul_mod(a1, b0, m), m.q);
    po[2 * pw] = mul_mod(a1, b1, m);
  }
}

// ---- BEHZ steps (6)-(8): scale by t, fast floor by q, Shenoy-Kumaresan back to q ----
// dq [polys][L][N], dB [polys][nBsk][N] (coefficient form) -> out [polys][L][N]
template <int LT, int NBT>
__global__ __launch_bounds__(256) void k_behz_floor(DevCtx c, const u64 *dq, const u64 *dB, u64 *out, size_t polys) {
  const ABC_CONST_AS DevConst &k = *(const ABC_CONST_AS DevConst *)c.cst;  // constants never change: scalar loads
  const int L = LT ? LT : k.nq, nB = LT ? NBT : k.nB, nBsk = nB + 1;
  const size_t items = polys * c.n;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const Mod msk = mod_at(c, c.id_bsk + nB);
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const size_t p = it >> c.logn, x = it & (c.n - 1);
    u64 tq[LT ? LT : kMaxLimbs], fl[LT ? NBT + 1 : kMaxLimbs + 1];
#pragma unroll
    for (int i = 0; i < L; i++) {
      // * t * (q/q_i)^-1; canonical: the conversion below sums these residues as INTEGERS in [0, q_i)
      tq[i] = mul_shoup(dq[(p * L + i) * c.n + x], k.flr_q[i], k.flr_q_s[i], mod_at(c, i).q);
    }
    u64 dep = tq[L - 1];
#pragma unroll
    for (int j = 0; j < nBsk; j++) {
      const Mod m = mod_at(c, c.id_bsk + j);
      const ABC_CONST_AS u64 *row = row_after(&k.q_to_bsk[j][0], dep), *row_s = row_after(&k.q_to_bsk_s[j][0], dep);
      const u64 conv = dot_shoup_lazy<LT>(L, [&](int i) { return tq[i]; }, [&](int i) { return row[i]; },
                                      [&](int i) { return row_s[i]; }, m);
      // (dB*t - conv) * q^-1 = dB*(t q^-1) - conv*q^-1   (both constants carry Shoup quotients)
      const u64 xb = mul_shoup(dB[(p * nBsk + j) * c.n + x], k.tinvq_bsk[j], k.tinvq_bsk_s[j], m.q);
      fl[j] = sub_mod(xb, mul_shoup(conv, k.inv_q_mod_bsk[j], k.inv_q_mod_bsk_s[j], m.q), m.q);
      dep = fl[j];
    }
    u64 tb[LT ? NBT : kMaxLimbs];
#pragma unroll
    for (int b2 = 0; b2 < nB; b2++) tb[b2] = mul_shoup(fl[b2], k.inv_punct_B[b2], k.inv_punct_B_s[b2], mod_at(c, c.id_bsk + b2).q);
    const u64 msk_conv = canon_dot(dot_shoup_lazy<NBT>(nB, [&](int b2) { return tb[b2]; }, [&](int b2) { return k.B_to_msk[b2]; },
                                                  [&](int b2) { return k.B_to_msk_s[b2]; }, msk), msk);
    const u64 alpha = mul_shoup(sub_mod(msk_conv, fl[nB], msk.q), k.inv_B_mod_msk, k.inv_B_mod_msk_s, msk.q);
    const bool neg = alpha > (msk.q >> 1);
    dep = alpha;
#pragma unroll
    for (int i = 0; i < L; i++) {
      const Mod m = mod_at(c, i);
      const ABC_CONST_AS u64 *row = row_after(&k.B_to_q[i][0], dep), *row_s = row_after(&k.B_to_q_s[i][0], dep);
      u64 v = dot_shoup_lazy<NBT>(nB, [&](int b2) { return tb[b2]; }, [&](int b2) { return row[b2]; },
                             [&](int b2) { return row_s[b2]; }, m);
      // +- alpha * B: one more lazy term (the subtraction as 2q - term keeps the sum non-negative)
      const u64 ab = mul_shoup_lazy(neg ? msk.q - alpha : alpha, k.B_mod_q[i], k.B_mod_q_s[i], m.q);
      v = neg ? v + ab : v + m.two_q - ab;
      if (m.bits > 58) v = csub(v, m.two_q);  // v < 2q + 2q there
      dep = canon_dot(v, m);
      out[(p * L + i) * c.n + x] = dep;
    }
  }
}


// ---- fp64 twins of the two BEHZ kernels (every ciphertext prime and every auxiliary prime below 2^50) ----
// Same steps, same integers: each modular product is fp_mul_lazy on an integer-valued double (exact while magnitudes stay
// below 2^53), sums of up to 16 lazily reduced terms are exact, and every value the algorithm uses AS AN INTEGER (the
// residues summed by the fast base conversions, r mod m~, alpha) is brought to its canonical representative first --
// 6 DP instructions per term of a conversion instead of 10 integer ones.
__device__ __forceinline__ double fp_canon_d(double x, double q, double qinv) {  // any lazy value -> canonical [0, q) as a double
  const double r = fp_centre(x, q, qinv);
  return r < 0.0 ? r + q : r;
}
template <int LT, int NBT>
__global__ __launch_bounds__(256) void k_behz_extend_fp(DevCtx c, const u64 *in, const u64 *in2, u64 *out, size_t polys) {
  const ABC_CONST_AS DevConst &k = *(const ABC_CONST_AS DevConst *)c.cst;
  const ABC_CONST_AS DevConstFp &f = *(const ABC_CONST_AS DevConstFp *)c.cstf;
  const int L = LT ? LT : k.nq, nBsk = LT ? NBT + 1 : k.nBsk;
  const size_t items = polys * c.n;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const size_t p = it >> c.logn, x = it & (c.n - 1);
    const bool second = in2 && p >= polys / 2;
    const u64 *src = second ? in2 : in;
    const size_t pp = second ? p - polys / 2 : p;
    double tmp[LT ? LT : kMaxLimbs];
    u32 mt = 0;
#pragma unroll
    for (int i = 0; i < L; i++) {
      const Mod m = mod_at(c, i);
      const double v = fp_canon_d(fp_mul_lazy(fp_from_u64(src[(pp * L + i) * c.n + x]), f.ext_q[i][0], f.ext_q[i][1], m.qd), m.qd, m.qinv);
      tmp[i] = v;
      mt += (u32)(u64)v * (u32)k.q_to_mtilde[i];  // arithmetic mod 2^32 on the canonical residue
    }
    const u32 r32 = mt * (u32)k.neg_inv_q_mod_mtilde;
    const double r = (double)(int)r32;  // centred representative of r mod m~ (r32 >= 2^31 -> r32 - 2^32)
    u64 dep = (u64)r32;  // keeps every iteration's constant loads inside the iteration (row_after)
#pragma unroll
    for (int j = 0; j < nBsk; j++) {
      const Mod m = mod_at(c, c.id_bsk + j);
      const ABC_CONST_AS double *row = row_after(&f.q_to_bsk[j][0][0], dep);
      double conv = fp_mul_lazy(r, f.q_mod_bsk[j][0], f.q_mod_bsk[j][1], m.qd);
#pragma unroll
      for (int i = 0; i < L; i++) conv += fp_mul_lazy(tmp[i], row[2 * i], row[2 * i + 1], m.qd);
      dep = fp_to_canon(fp_mul_lazy(conv, f.inv_mtilde_mod_bsk[j][0], f.inv_mtilde_mod_bsk[j][1], m.qd), m.qd, m.qinv);
      out[(p * nBsk + j) * c.n + x] = dep;
    }
  }
}
template <int LT, int NBT>
__global__ __launch_bounds__(256) void k_behz_floor_fp(DevCtx c, const u64 *dq, const u64 *dB, u64 *out, size_t polys) {
  const ABC_CONST_AS DevConst &k = *(const ABC_CONST_AS DevConst *)c.cst;
  const ABC_CONST_AS DevConstFp &f = *(const ABC_CONST_AS DevConstFp *)c.cstf;
  const int L = LT ? LT : k.nq, nB = LT ? NBT : k.nB, nBsk = nB + 1;
  const size_t items = polys * c.n;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const Mod msk = mod_at(c, c.id_bsk + nB);
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const size_t p = it >> c.logn, x = it & (c.n - 1);
    double tq[LT ? LT : kMaxLimbs], fl[LT ? NBT + 1 : kMaxLimbs + 1];
#pragma unroll
    for (int i = 0; i < L; i++) {
      const Mod m = mod_at(c, i);
      tq[i] = fp_canon_d(fp_mul_lazy(fp_from_u64(dq[(p * L + i) * c.n + x]), f.flr_q[i][0], f.flr_q[i][1], m.qd), m.qd, m.qinv);
    }
    u64 dep = (u64)__double_as_longlong(tq[L - 1]);
#pragma unroll
    for (int j = 0; j < nBsk; j++) {
      const Mod m = mod_at(c, c.id_bsk + j);
      const ABC_CONST_AS double *row = row_after(&f.q_to_bsk[j][0][0], dep);
      double conv = 0.0;
#pragma unroll
      for (int i = 0; i < L; i++) conv += fp_mul_lazy(tq[i], row[2 * i], row[2 * i + 1], m.qd);
      // (dB*t - conv) * q^-1 = dB*(t q^-1) - conv*q^-1
      const double xb = fp_mul_lazy(fp_from_u64(dB[(p * nBsk + j) * c.n + x]), f.tinvq_bsk[j][0], f.tinvq_bsk[j][1], m.qd);
      fl[j] = xb - fp_mul_lazy(conv, f.inv_q_mod_bsk[j][0], f.inv_q_mod_bsk[j][1], m.qd);
      dep = (u64)__double_as_longlong(fl[j]);
    }
    double tb[LT ? NBT : kMaxLimbs];
#pragma unroll
    for (int b2 = 0; b2 < nB; b2++) {
      const Mod m = mod_at(c, c.id_bsk + b2);
      tb[b2] = fp_canon_d(fp_mul_lazy(fl[b2], f.inv_punct_B[b2][0], f.inv_punct_B[b2][1], m.qd), m.qd, m.qinv);
    }
    double mconv = -fl[nB];
#pragma unroll
    for (int b2 = 0; b2 < nB; b2++) mconv += fp_mul_lazy(tb[b2], f.B_to_msk[b2][0], f.B_to_msk[b2][1], msk.qd);
    // alpha = (conv - x_msk) B^-1 mod m_sk, as the signed integer in [-(m_sk-1)/2, (m_sk-1)/2] Shenoy-Kumaresan subtracts
    double alpha = fp_canon_d(fp_mul_lazy(mconv, f.inv_B_mod_msk[0], f.inv_B_mod_msk[1], msk.qd), msk.qd, msk.qinv);
    if (alpha > (double)(msk.q >> 1)) alpha -= msk.qd;
    dep = (u64)__double_as_longlong(alpha);
#pragma unroll
    for (int i = 0; i < L; i++) {
      const Mod m = mod_at(c, i);
      const ABC_CONST_AS double *row = row_after(&f.B_to_q[i][0][0], dep);
      double v = -fp_mul_lazy(alpha, f.B_mod_q[i][0], f.B_mod_q[i][1], m.qd);
#pragma unroll
      for (int b2 = 0; b2 < nB; b2++) v += fp_mul_lazy(tb[b2], row[2 * b2], row[2 * b2 + 1], m.qd);
      dep = fp_to_canon(v, m.qd, m.qinv);
      out[(p * L + i) * c.n + x] = dep;
    }
  }
}

// ---- BEHZ steps (4)-(5) for rings that fit LDS: dyadic tensor product fused into the load of the inverse transform ----
// workgroup (ct, comp, limb): d_comp = a0 b0 | a0 b1 + a1 b0 | a1 b1 of limb `limb`, inverse transform in LDS, coefficient
// form to d [count][3][nlm][N].  Saves the tensor kernels' round trip (7 limb transfers per limb and ciphertext).
template <int LB, bool FP>
__global__ __launch_bounds__((1 << LB) / 16) void k_bfv_tensor_intt(DevCtx c, const u64 *__restrict__ a, const u64 *__restrict__ b,
                                                                    u64 *__restrict__ d, LimbMap map, int nlm) {
  __shared__ u64 lds_raw[lds_words(LB)];
  const size_t N = (size_t)1 << LB, pw = (size_t)nlm * N;
  const int limb = blockIdx.x % nlm;
  const int comp = (blockIdx.x / nlm) % 3;
  const size_t ct = blockIdx.x / (3 * (size_t)nlm);
  const int mid = map.id[limb];
  const Mod m = mod_at(c, mid);
  const u64 *__restrict__ a0 = a + ct * 2 * pw + (size_t)limb * N, *__restrict__ a1 = a0 + pw;
  const u64 *__restrict__ b0 = b + ct * 2 * pw + (size_t)limb * N, *__restrict__ b1 = b0 + pw;
  u64 *__restrict__ o = d + (ct * 3 + comp) * pw + (size_t)limb * N;
  if constexpr (FP) {
    double *lds = reinterpret_cast<double *>(lds_raw);
    const FpTable t = fp_table(c, mid);
    const double q = m.qd, qinv = m.qinv;
    auto st = [&](int, int i, double v) { o[i] = fp_to_canon(fp_mul_lazy(v, m.inv_n_c, m.inv_n_cq, q), q, qinv); };
    auto prod = [&](double x, double y) {  // |x|, |y| <= q  ->  |result| < q
      const double h = x * y, l = __builtin_fma(x, y, -h);
      return __builtin_fma(-__builtin_rint(h * qinv), q, h) + l;
    };
    if (comp == 0)  // workgroup-uniform: three straight-line bodies
      ntt_inv_block_a<LB, FpArith>(lds, [&](int, int i) { return prod(fp_from_u64(a0[i]), fp_from_u64(b0[i])); }, st, t, m, 0, 0);
    else if (comp == 2)
      ntt_inv_block_a<LB, FpArith>(lds, [&](int, int i) { return prod(fp_from_u64(a1[i]), fp_from_u64(b1[i])); }, st, t, m, 0, 0);
    else
      ntt_inv_block_a<LB, FpArith>(
          lds,
          [&](int, int i) {  // the sum of two products is re-centred: the first inverse pass starts from |x| <= q/2
            return fp_centre(prod(fp_from_u64(a0[i]), fp_from_u64(b1[i])) + prod(fp_from_u64(a1[i]), fp_from_u64(b0[i])), q, qinv);
          },
          st, t, m, 0, 0);
  } else {
    const NttTable t = ntt_table(c, mid);
    auto st = [&](int, int i, u64 v) { o[i] = scale_inv_n(v, m); };
    if (comp == 0)
      ntt_inv_block<LB>(lds_raw, [&](int, int i) { return mul_mod(a0[i], b0[i], m); }, st, t, m, 0, 0);
    else if (comp == 2)
      ntt_inv_block<LB>(lds_raw, [&](int, int i) { return mul_mod(a1[i], b1[i], m); }, st, t, m, 0, 0);
    else
      ntt_inv_block<LB>(
          lds_raw,
          [&](int, int i) {
            U128 acc = mul_wide(a0[i], b1[i]);
            mac128(acc, a1[i], b0[i]);
            return barrett_reduce(acc, m);
          },
          st, t, m, 0, 0);
  }
}

// The same fusion for N = 2^15 / 2^16, where a limb is 2^S0 blocks of 2^LB points behind a strided pass: workgroup
// (ct, comp, limb, block) forms its block of the product and runs the block stages of the inverse transform; the strided pass
// (launch_ntt_inv_strided_part) finishes.  Replaces k_tensor_map + the block kernel of launch_ntt_inv: 3 limb transfers per
// product limb fewer.
template <int LB, bool FP>
__global__ __launch_bounds__((1 << LB) / 16) void k_bfv_tensor_inv_block(DevCtx c, const u64 *__restrict__ a, const u64 *__restrict__ b,
                                                                         u64 *__restrict__ d, LimbMap map, int nlm, int S0,
                                                                         size_t ct_stride) {
  // ct_stride: words between the operands of consecutive ciphertexts (2 nlm N where a and b are separate arrays of size-2
  // ciphertexts; 4 nlm N where the four operand polynomials of a pair lie together: abc_kernels_bmul.hip)
  __shared__ u64 lds_raw[lds_words(LB)];
  const size_t N = (size_t)c.n, pw = (size_t)nlm * N;
  const int blk = blockIdx.x & ((1 << S0) - 1);
  const unsigned w = blockIdx.x >> S0;
  const int limb = w % nlm;
  const int comp = (w / nlm) % 3;
  const size_t ct = w / (3 * (size_t)nlm);
  const int mid = map.id[limb];
  const Mod m = mod_at(c, mid);
  const size_t boff = (size_t)limb * N + ((size_t)blk << LB);
  const u64 *__restrict__ a0 = a + ct * ct_stride + boff, *__restrict__ a1 = a0 + pw;
  const u64 *__restrict__ b0 = b + ct * ct_stride + boff, *__restrict__ b1 = b0 + pw;
  u64 *__restrict__ o = d + (ct * 3 + comp) * pw + boff;
  if constexpr (FP) {
    double *lds = reinterpret_cast<double *>(lds_raw);
    const FpTable t = fp_table(c, mid);
    const double q = m.qd, qinv = m.qinv;
    auto st = [&](int, int i, double v) { reinterpret_cast<double *>(o)[i] = v; };  // raw doubles: the strided pass re-centres
    auto prod = [&](double x, double y) {
      const double h = x * y, l = __builtin_fma(x, y, -h);
      return __builtin_fma(-__builtin_rint(h * qinv), q, h) + l;
    };
    if (comp == 0)
      ntt_inv_block_a<LB, FpArith>(lds, [&](int, int i) { return prod(fp_from_u64(a0[i]), fp_from_u64(b0[i])); }, st, t, m, S0, blk);
    else if (comp == 2)
      ntt_inv_block_a<LB, FpArith>(lds, [&](int, int i) { return prod(fp_from_u64(a1[i]), fp_from_u64(b1[i])); }, st, t, m, S0, blk);
    else
      ntt_inv_block_a<LB, FpArith>(
          lds,
          [&](int, int i) {
            return fp_centre(prod(fp_from_u64(a0[i]), fp_from_u64(b1[i])) + prod(fp_from_u64(a1[i]), fp_from_u64(b0[i])), q, qinv);
          },
          st, t, m, S0, blk);
  } else {
    const NttTable t = ntt_table(c, mid);
    auto st = [&](int, int i, u64 v) { o[i] = v; };  // [0, 2q): the strided pass continues from there
    if (comp == 0)
      ntt_inv_block<LB>(lds_raw, [&](int, int i) { return mul_mod(a0[i], b0[i], m); }, st, t, m, S0, blk);
    else if (comp == 2)
      ntt_inv_block<LB>(lds_raw, [&](int, int i) { return mul_mod(a1[i], b1[i], m); }, st, t, m, S0, blk);
    else
      ntt_inv_block<LB>(
          lds_raw,
          [&](int, int i) {
            U128 acc = mul_wide(a0[i], b1[i]);
            mac128(acc, a1[i], b0[i]);
            return barrett_reduce(acc, m);
          },
          st, t, m, S0, blk);
  }
}
// 0 done, -1 not applicable, 1 error
static int tensor_inv_big(abc_hip_ctx *c, const u64 *a, const u64 *b, u64 *d, const LimbMap &map, int nlm, size_t count) {
  if ((c->logn != 15 && c->logn != 16) || c->sw.no_tensor_intt || big_block_log() != 12) return -1;
  bool fp = c->use_fp;
  for (int j = 0; j < nlm; j++) fp = fp && fp_ok(c->h_mods[map.id[j]].bits);
  const int S0 = c->logn - 12;
  const dim3 grid((unsigned)((count * 3 * nlm) << S0)), block((1 << 12) / 16);
  const size_t cts = 2 * (size_t)nlm * c->n;
  if (fp)
    hipLaunchKernelGGL((k_bfv_tensor_inv_block<12, true>), grid, block, 0, c->stream, c->dc, a, b, d, map, nlm, S0, cts);
  else
    hipLaunchKernelGGL((k_bfv_tensor_inv_block<12, false>), grid, block, 0, c->stream, c->dc, a, b, d, map, nlm, S0, cts);
  ABC_HIP_CHECK(hipGetLastError());
  return launch_ntt_inv_strided_part(c, d, map, nlm, count * 3 * nlm);
}
// the fp64 form alone, operands at a caller-given ciphertext stride, no strided pass behind it (abc_kernels_bmul.hip finishes)
int launch_bfv_tensor_inv_block(abc_hip_ctx *c, const u64 *a, const u64 *b, size_t ct_stride, u64 *d, const LimbMap &map, int nlm,
                                size_t count) {
  if ((c->logn != 15 && c->logn != 16) || big_block_log() != 12) { set_error("tensor_inv_block: N = 2^15 / 2^16 only"); return 1; }
  const int S0 = c->logn - 12;
  const dim3 grid((unsigned)((count * 3 * nlm) << S0)), block((1 << 12) / 16);
  hipLaunchKernelGGL((k_bfv_tensor_inv_block<12, true>), grid, block, 0, c->stream, c->dc, a, b, d, map, nlm, S0, ct_stride);
  ABC_HIP_CHECK(hipGetLastError());
  return 0;
}

template <int LB>
static int launch_tensor_intt(abc_hip_ctx *c, const u64 *a, const u64 *b, u64 *d, const LimbMap &map, int nlm, size_t count) {
  bool fp = c->use_fp;
  for (int j = 0; j < nlm; j++) fp = fp && fp_ok(c->h_mods[map.id[j]].bits);
  const dim3 grid((unsigned)(count * 3 * nlm)), block((1 << LB) / 16);
  if (fp)
    hipLaunchKernelGGL((k_bfv_tensor_intt<LB, true>), grid, block, 0, c->stream, c->dc, a, b, d, map, nlm);
  else
    hipLaunchKernelGGL((k_bfv_tensor_intt<LB, false>), grid, block, 0, c->stream, c->dc, a, b, d, map, nlm);
  ABC_HIP_CHECK(hipGetLastError());
  return 0;
}
// -1: ring too large for an LDS-resident limb
static int tensor_intt(abc_hip_ctx *c, const u64 *a, const u64 *b, u64 *d, const LimbMap &map, int nlm, size_t count) {
  if (c->sw.no_tensor_intt) return -1;
  // single-ciphertext calls: the separate kernels spread a transform over many workgroups (launch_ntt, "few limbs")
  if (c->logn == 14 && count * 3 * (size_t)nlm <= 48) return -1;
  switch (c->logn) {
    case 10: return launch_tensor_intt<10>(c, a, b, d, map, nlm, count);
    case 11: return launch_tensor_intt<11>(c, a, b, d, map, nlm, count);
    case 12: return launch_tensor_intt<12>(c, a, b, d, map, nlm, count);
    case 13: return launch_tensor_intt<13>(c, a, b, d, map, nlm, count);
    case 14: return launch_tensor_intt<14>(c, a, b, d, map, nlm, count);
    default: return -1;
  }
}

// shapes of BFVDefault(4096 / 8192 / 16384) and of config 5 get fully unrolled kernels (the L = 8, nB = 9 shape was tried:
// slower than the generic loop form, 658 against 740 circuits/s on config 5 with a 50-bit chain)
#define ABC_BEHZ_DISPATCH(KERNEL, GRID, ...)                                                                          \
  do {                                                                                                                  \
    const int L_ = c->L, nB_ = c->nB;                                                                                   \
    if (L_ == 2 && nB_ == 2) hipLaunchKernelGGL((KERNEL<2, 2>), GRID, dim3(256), 0, c->stream, __VA_ARGS__);            \
    else if (L_ == 4 && nB_ == 4) hipLaunchKernelGGL((KERNEL<4, 4>), GRID, dim3(256), 0, c->stream, __VA_ARGS__);       \
    else if (L_ == 8 && nB_ == 8) hipLaunchKernelGGL((KERNEL<8, 8>), GRID, dim3(256), 0, c->stream, __VA_ARGS__);       \
    else hipLaunchKernelGGL((KERNEL<0, 0>), GRID, dim3(256), 0, c->stream, __VA_ARGS__);                                \
  } while (0)
static void launch_behz_extend(abc_hip_ctx *c, const u64 *in, const u64 *in2, u64 *out, size_t polys) {
  if (c->behz_fp) {
    ABC_BEHZ_DISPATCH(k_behz_extend_fp, dim3(grid_for(polys * c->n, 256)), c->dc, in, in2, out, polys);
    return;
  }
  ABC_BEHZ_DISPATCH(k_behz_extend, dim3(grid_for(polys * c->n, 256)), c->dc, in, in2, out, polys);
}
static void launch_behz_floor(abc_hip_ctx *c, const u64 *dq, const u64 *dB, u64 *out, size_t polys) {
  if (c->behz_fp) {
    ABC_BEHZ_DISPATCH(k_behz_floor_fp, dim3(grid_for(polys * c->n, 256)), c->dc, dq, dB, out, polys);
    return;
  }
  ABC_BEHZ_DISPATCH(k_behz_floor, dim3(grid_for(polys * c->n, 256)), c->dc, dq, dB, out, polys);
}

int bfv_multiply(abc_hip_ctx *c, const u64 *a, const u64 *b, u64 *out3, size_t count) {
  if (c->scheme != 1) { set_error("bfv_multiply on a non-BFV context"); return 1; }
  if (!count) return 0;
  if (bmul_multiply_applies(c)) return bmul_split(c, a, b, out3, count, false);  // BFVDefault shape on fp64 primes: abc_kernels_bmul.hip
  const size_t N = (size_t)c->n;
  const int L = c->L, nBsk = c->nBsk, nlm = L + nBsk;
  // per ciphertext pair (words): aq,bq 2*2L ; aB,bB 2*2nBsk ; dq 3L ; dB 3nBsk
  const size_t per_ct = (size_t)(4 * L + 4 * nBsk + 3 * L + 3 * nBsk) * N;
  size_t budget = (size_t)4 << 30;  // scratch capped at 4 GiB (ABC_HIP_BFV_SCRATCH_MB: test knob, forces several chunks)
  if (c->sw.bfv_scratch_mb) budget = c->sw.bfv_scratch_mb << 20;
  size_t chunk = (budget / 8) / per_ct;
  if (chunk < 1) chunk = 1;
  if (chunk > count) chunk = count;
  else if (count % chunk && count / chunk < 8) chunk = (count + count / chunk) / (count / chunk + 1);  // even chunks, no runt
  if (ensure_workspace(c, chunk * per_ct * 8)) return 1;
  u64 *aq = (u64 *)c->ws;
  u64 *aB = aq + chunk * 4 * L * N;
  u64 *dq = aB + chunk * 4 * nBsk * N, *dB = dq + chunk * 3 * L * N;
  const LimbMap qmap = key_limb_map(c, L);
  LimbMap bmap{};
  for (int j = 0; j < nBsk; j++) bmap.id[j] = c->dc.id_bsk + j;
  (void)nlm;
  for (size_t off = 0; off < count; off += chunk) {
    const size_t cc = (count - off < chunk) ? count - off : chunk;
    const u64 *pa = a + off * 2 * L * N, *pb = b + off * 2 * L * N;
    // both operands go through one launch each of the extension and of the transforms, which write [a polys | b polys]
    // back to back: the b halves therefore start cc (not chunk) ciphertexts behind the a halves.  (Round 1 used the fixed
    // chunk offset here, which is only the same thing when the last chunk is full: a ragged last chunk -- 125 pairs at
    // N = 2^16 split 63 + 62 -- read stale b operands.  Found by running config 5 at its stated per-GPU batch.)
    u64 *bq = aq + cc * 2 * L * N, *bB = aB + cc * 2 * nBsk * N;
    launch_behz_extend(c, pa, pb, aB, cc * 4);
    ABC_HIP_CHECK(hipGetLastError());
    // operands stay intact: transform out of place, both operands in one launch (aq and bq are adjacent)
    if (launch_ntt_fwd_from2(c, pa, pb, aq, qmap, L, cc * 4 * L)) return 1;
    if (launch_ntt_fwd(c, aB, bmap, nBsk, cc * 4 * nBsk)) return 1;    // aB and bB are adjacent
    int rq = tensor_intt(c, aq, bq, dq, qmap, L, cc);
    if (rq < 0) rq = tensor_inv_big(c, aq, bq, dq, qmap, L, cc);
    if (rq > 0) return 1;
    if (rq < 0) {
      hipLaunchKernelGGL(k_tensor_map, dim3(grid_for(cc * L * N, 256)), dim3(256), 0, c->stream, c->dc, aq, bq, dq, qmap, L, cc);
      ABC_HIP_CHECK(hipGetLastError());
      if (launch_ntt_inv(c, dq, qmap, L, cc * 3 * L)) return 1;
    }
    int rb = tensor_intt(c, aB, bB, dB, bmap, nBsk, cc);
    if (rb < 0) rb = tensor_inv_big(c, aB, bB, dB, bmap, nBsk, cc);
    if (rb > 0) return 1;
    if (rb < 0) {
      hipLaunchKernelGGL(k_tensor_map, dim3(grid_for(cc * nBsk * N, 256)), dim3(256), 0, c->stream, c->dc, aB, bB, dB, bmap, nBsk,
                         cc);
      ABC_HIP_CHECK(hipGetLastError());
      if (launch_ntt_inv(c, dB, bmap, nBsk, cc * 3 * nBsk)) return 1;
    }
    launch_behz_floor(c, dq, dB, out3 + off * 3 * L * N, cc * 3);
    ABC_HIP_CHECK(hipGetLastError());
  }
  return 0;
}

// ---- multiply_plain: lift plaintext to each q_i, NTT, dyadic multiply, INTT ----
__global__ __launch_bounds__(256) void k_plain_lift(DevCtx c, const u64 *plain, u64 *lifted, size_t count) {
  const DevConst &k = *c.cst;
  const size_t items = count * c.n;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const size_t p = it >> c.logn, x = it & (c.n - 1);
    const u64 v = plain[it];
    const bool upper = v >= k.upper_half_threshold;
    for (int i = 0; i < c.L; i++) lifted[(p * c.L + i) * c.n + x] = v + (upper ? k.upper_half_increment[i] : 0);
  }
}
__global__ __launch_bounds__(256) void k_mul_plain_ntt(DevCtx c, u64 *ct, const u64 *lifted, size_t lifted_stride, int size,
                                                       size_t count) {
  const size_t pw = (size_t)c.L * c.n;
  const size_t items = count * size * pw;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const size_t ctp = it / pw, w = it % pw;
    const size_t ci = ctp / size;
    ct[it] = mul_mod(ct[it], lifted[ci * lifted_stride + w], c.mods[w >> c.logn]);
  }
}

// multiply_plain for rings that fit LDS and primes below 2^50, one kernel per ciphertext limb: forward transform in LDS,
// dyadic product with the plaintext's NTT form in registers (the forward transform's last pass leaves each lane exactly the
// words the inverse transform's first pass wants), inverse transform in LDS.  3 limb transfers instead of 9.
template <int LB>
__global__ __launch_bounds__((1 << LB) / 16) void k_bfv_mul_plain_fused_fp(DevCtx c, const u64 *__restrict__ ct,
                                                                           const u64 *__restrict__ lifted, size_t lifted_stride,
                                                                           u64 *__restrict__ out, int size) {
  __shared__ double lds[lds_words(LB)];
  const size_t N = (size_t)1 << LB;
  const int L = c.L;
  const int limb = blockIdx.x % L;
  const size_t poly = blockIdx.x / L;  // ct * size + comp
  const size_t ci = poly / size;
  const Mod m = mod_at(c, limb);
  const FpTable t = fp_table(c, limb);
  const double q = m.qd, qinv = m.qinv;
  const u64 *__restrict__ src = ct + (poly * L + limb) * N;
  const u64 *__restrict__ pl = lifted + ci * lifted_stride + (size_t)limb * N;
  u64 *__restrict__ dst = out + (poly * L + limb) * N;
  double y[16];
  ntt_fwd_block_a<LB, FpArith>(
      lds, [&](int, int i) { return fp_from_u64(src[i]); },
      [&](int r, int i, double v) {
        const double w = fp_from_u64(pl[i]);
        const double h = v * w, l = __builtin_fma(v, w, -h);
        y[r] = fp_centre(__builtin_fma(-__builtin_rint(h * qinv), q, h) + l, q, qinv);
      },
      t, m, 0, 0);
  block_sync_lds();  // the forward transform's last pass has read its LDS words
  ntt_inv_block_a<LB, FpArith>(
      lds, [&](int r, int) { return y[r]; },
      [&](int, int i, double v) { dst[i] = fp_to_canon(fp_mul_lazy(v, m.inv_n_c, m.inv_n_cq, q), q, qinv); }, t, m, 0, 0);
}
template <int LB>
static int launch_mul_plain_fused(abc_hip_ctx *c, const u64 *ct, const u64 *lifted, size_t lifted_stride, u64 *out, int size,
                                  size_t count) {
  hipLaunchKernelGGL(k_bfv_mul_plain_fused_fp<LB>, dim3((unsigned)(count * size * c->L)), dim3((1 << LB) / 16), 0, c->stream, c->dc,
                     ct, lifted, lifted_stride, out, size);
  ABC_HIP_CHECK(hipGetLastError());
  return 0;
}

int bfv_multiply_plain(abc_hip_ctx *c, const u64 *ct, const u64 *plain, size_t plain_stride, u64 *out, int size, size_t count) {
  if (!count) return 0;
  const size_t N = (size_t)c->n;
  const int L = c->L;
  const size_t nplain = plain_stride ? count : 1;
  if (ensure_workspace(c, nplain * L * N * 8)) return 1;
  u64 *lifted = (u64 *)c->ws;
  const LimbMap qmap = key_limb_map(c, L);
  if (plain_stride && plain_stride != N) { set_error("multiply_plain: plain_stride must be 0 or N"); return 1; }
  hipLaunchKernelGGL(k_plain_lift, dim3(grid_for(nplain * N, 256)), dim3(256), 0, c->stream, c->dc, plain, lifted, nplain);
  ABC_HIP_CHECK(hipGetLastError());
  if (launch_ntt_fwd(c, lifted, qmap, L, nplain * L)) return 1;
  bool fp = c->use_fp && c->logn <= 14 && !c->sw.no_fused;
  for (int j = 0; j < L; j++) fp = fp && fp_ok(c->h_mods[j].bits);
  if (fp && (c->logn < 14 || count * size * L > 48)) {  // single-ciphertext calls keep the spread-out transforms
    const size_t ls = plain_stride ? (size_t)L * N : 0;
    switch (c->logn) {
      case 10: return launch_mul_plain_fused<10>(c, ct, lifted, ls, out, size, count);
      case 11: return launch_mul_plain_fused<11>(c, ct, lifted, ls, out, size, count);
      case 12: return launch_mul_plain_fused<12>(c, ct, lifted, ls, out, size, count);
      case 13: return launch_mul_plain_fused<13>(c, ct, lifted, ls, out, size, count);
      case 14: return launch_mul_plain_fused<14>(c, ct, lifted, ls, out, size, count);
      default: break;
    }
  }
  if (out != ct) ABC_HIP_CHECK(hipMemcpyAsync(out, ct, count * size * L * N * 8, hipMemcpyDeviceToDevice, c->stream));
  if (launch_ntt_fwd(c, out, qmap, L, count * size * L)) return 1;
  hipLaunchKernelGGL(k_mul_plain_ntt, dim3(grid_for(count * size * L * N, 256)), dim3(256), 0, c->stream, c->dc, out, lifted,
                     plain_stride ? (size_t)L * N : 0, size, count);
  ABC_HIP_CHECK(hipGetLastError());
  return launch_ntt_inv(c, out, qmap, L, count * size * L);
}

// ---- add_plain / sub_plain: c0 +/- round(q*m/t)  (multiply_add_plain_with_scaling_variant) ----
__global__ __launch_bounds__(256) void k_plain_scale_addsub(DevCtx c, const u64 *ct, const u64 *plain, size_t plain_stride, u64 *out,
                                                            int size, size_t count, int sub) {
  const DevConst &k = *c.cst;
  const Mod mt = c.mods[c.id_t];
  const size_t items = count * c.n;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t pw = (size_t)c.L * c.n;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const size_t ci = it >> c.logn, x = it & (c.n - 1);
    const u64 m = plain[ci * plain_stride + x];
    // fix = floor((m * (q mod t) + (t+1)/2) / t); numerator < t^2 + t < 2^41
    const u64 numer = m * k.q_mod_t + k.upper_half_threshold;
    const u64 rem = reduce64(numer, mt);
    const u64 fix = (numer - rem) / mt.q;  // exact division of a < 2^41 value
    for (int i = 0; i < c.L; i++) {
      const Mod mq = c.mods[i];
      const u64 scaled = add_mod(mul_mod(m, k.coeff_div_plain[i], mq), fix, mq.q);
      const size_t o = ci * size * pw + (size_t)i * c.n + x;
      out[o] = sub ? sub_mod(ct[o], scaled, mq.q) : add_mod(ct[o], scaled, mq.q);
    }
  }
}

int bfv_addsub_plain(abc_hip_ctx *c, const u64 *ct, const u64 *plain, size_t plain_stride, u64 *out, int size, size_t count,
                     int sub) {
  if (!count) return 0;
  const size_t N = (size_t)c->n;
  if (out != ct) ABC_HIP_CHECK(hipMemcpyAsync(out, ct, count * size * c->L * N * 8, hipMemcpyDeviceToDevice, c->stream));
  hipLaunchKernelGGL(k_plain_scale_addsub, dim3(grid_for(count * N, 256)), dim3(256), 0, c->stream, c->dc, out, plain,
                     plain_stride, out, size, count, sub);
  ABC_HIP_CHECK(hipGetLastError());
  return 0;
}

// ---- BatchEncoder ----
__global__ __launch_bounds__(256) void k_batch_scatter(DevCtx c, const int64_t *values, u64 *plain, size_t count) {
  const u64 t = c.cst->t;
  const size_t items = count * c.n;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const size_t p = it >> c.logn, i = it & (c.n - 1);
    const int64_t v = values[it];
    plain[p * c.n + c.slot_map[i]] = v < 0 ? t + (u64)v : (u64)v;
  }
}
__global__ __launch_bounds__(256) void k_batch_gather(DevCtx c, const u64 *tmp, int64_t *values, size_t count) {
  const u64 t = c.cst->t, half = t >> 1;
  const size_t items = count * c.n;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const size_t p = it >> c.logn, i = it & (c.n - 1);
    const u64 v = tmp[p * c.n + c.slot_map[i]];
    values[it] = v > half ? (int64_t)v - (int64_t)t : (int64_t)v;
  }
}

int batch_encode(abc_hip_ctx *c, const int64_t *values, u64 *plain, size_t count) {
  if (c->scheme != 1) { set_error("batch_encode needs a BFV context"); return 1; }
  if (!count) return 0;
  hipLaunchKernelGGL(k_batch_scatter, dim3(grid_for(count * c->n, 256)), dim3(256), 0, c->stream, c->dc, values, plain, count);
  ABC_HIP_CHECK(hipGetLastError());
  LimbMap tmap{};
  tmap.id[0] = c->dc.id_t;
  return launch_ntt_inv(c, plain, tmap, 1, count);
}

int batch_decode(abc_hip_ctx *c, const u64 *plain, int64_t *values, size_t count) {
  if (c->scheme != 1) { set_error("batch_decode needs a BFV context"); return 1; }
  if (!count) return 0;
  const size_t N = (size_t)c->n;
  if (ensure_workspace(c, count * N * 8)) return 1;
  u64 *tmp = (u64 *)c->ws;
  ABC_HIP_CHECK(hipMemcpyAsync(tmp, plain, count * N * 8, hipMemcpyDeviceToDevice, c->stream));
  LimbMap tmap{};
  tmap.id[0] = c->dc.id_t;
  if (launch_ntt_fwd(c, tmp, tmap, 1, count)) return 1;
  hipLaunchKernelGGL(k_batch_gather, dim3(grid_for(count * N, 256)), dim3(256), 0, c->stream, c->dc, tmp, values, count);
  ABC_HIP_CHECK(hipGetLastError());
  return 0;
}

// ---- BFV decryption rounding: phase [count][L][N] (coefficient form) -> plain [count][N] ----
__global__ __launch_bounds__(256) void k_bfv_decrypt_round(DevCtx c, const u64 *phase, u64 *plain, size_t count) {
  const DevConst &k = *c.cst;
  const Mod mt = c.mods[c.id_t], mg = c.mods[c.id_gamma];
  const size_t items = count * c.n;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const size_t p = it >> c.logn, x = it & (c.n - 1);
    u64 tmp[kMaxLimbs];
    for (int i = 0; i < c.L; i++) {
      tmp[i] = mul_shoup(phase[(p * c.L + i) * c.n + x], k.dec_q[i], k.dec_q_s[i], c.mods[i].q);  // * t*gamma*(q/q_i)^-1
    }
    u64 vt = dot_mod(c.L, [&](int i) { return tmp[i]; }, [&](int i) { return k.q_to_t[i]; }, mt);
    u64 vg = dot_mod(c.L, [&](int i) { return tmp[i]; }, [&](int i) { return k.q_to_gamma[i]; }, mg);
    vt = mul_mod(vt, k.neg_inv_q_mod_t, mt);
    vg = mul_mod(vg, k.neg_inv_q_mod_gamma, mg);
    u64 r;
    if (vg > (mg.q >> 1)) r = add_mod(vt, reduce64(mg.q - vg, mt), mt.q);
    else r = sub_mod(vt, reduce64(vg, mt), mt.q);
    if (r) r = mul_mod(r, k.inv_gamma_mod_t, mt);
    plain[it] = r;
  }
}

int launch_bfv_decrypt_round(abc_hip_ctx *c, const u64 *phase, u64 *plain, size_t count) {
  hipLaunchKernelGGL(k_bfv_decrypt_round, dim3(grid_for(count * c->n, 256)), dim3(256), 0, c->stream, c->dc, phase, plain, count);
  ABC_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace abc
