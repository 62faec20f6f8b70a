// abc_kernels_fused.hip -- the hot path for N <= 2^14 (one workgroup per RNS limb, the limb resident in LDS):
// hybrid key switching for relinearisation and Galois rotations (BFV and CKKS) and the CKKS
// ciphertext x ciphertext multiply + relinearise front end.
//
// Replaces, for rings that fit LDS:
//   SealCiphertext::multiply / multiplyInplace = Evaluator::multiply + relinearize_inplace
//       (src/runtime/SealCiphertext.cpp:102-107,121-124)                     -> ckks_mul_relin_fused, keyswitch_fused
//   SealCiphertext::rotateRows / rotateRowsInplace = Evaluator::rotate_rows    (:52-61) -> keyswitch_fused
// Three launch sequences live here; the dispatcher (run_mul_relin / run_keyswitch / keyswitch_stage) takes the first one the
// context's primes and ring size allow, and tests/test_gpu_paths.py holds all of them bit-identical to the oracle:
//   split fp64  (N = 2^14, every key prime < 2^50; the headline configuration; sequence described above k_split2_tensor_pass0_fp)
//       K1  k_split2_tensor_pass0_fp / k_fused_operand_pass0_fp : tensor product or operand, inverse transform of limb j in LDS,
//                                          first radix-16 register pass of the forward transforms modulo the other key primes;
//                                          half-done limbs to scratch, packed (abc_ntt.hpp) or as raw doubles
//       K2a k_split_special_fp, K2b k_split3_pass_fp, K2c k_split4_main_fp (k_split3_main_fp for six and seven limbs)
//   LDS-resident fp64 (N < 2^14, or ABC_HIP_NO_SPLIT; every key prime < 2^50)
//       K1 tensor_decomp_fp (operand_intt_fp + decomp_ntt_fp for a key switch), K2b mac, K2c special_intt_fp, K3 moddown_fp / _bfv_fp
//   LDS-resident integer (any prime up to 61 bits)
//       K1  tensor_intt : c0 = a0b0, c1 = a0b1 + a1b0 to scratch (so `out` may alias an operand); c2 = a1b1 kept in
//                         NTT form and, through an in-LDS inverse transform, in coefficient form
//       K2a decomp_ntt  : workgroup (ct, key prime I, limb J) reduces operand limb J modulo key prime I and
//                         transforms it in LDS
//       K2b mac         : streaming inner product with key[J][.][I] (128-bit lazy accumulation)
//       K2c special     : the special-prime limb goes back to coefficients, plus the q_sp/2 rounding offset
//       K3  moddown     : workgroup (ct, comp, j): CKKS: reduce the special-prime polynomial modulo q_j, transform
//                         it, subtract, scale by q_sp^-1, add c0 / c1.  BFV: inverse-transform the accumulated limb,
//                         then the same subtract / scale / add in coefficient form
// (chains that mix wide and fp64-capable primes: abc_kernels_isplit.hip; N = 2^15: abc_kernels_gsplit.hip; BFV multiply:
// abc_kernels_bmul.hip)
// Algorithmic HBM bytes per multiply: 8N(6L + 2L(L+1)); measured: DESIGN.md section 4.
#include <type_traits>

#include "abc_context.hpp"
#include "abc_host_math.hpp"

namespace abc {

static inline unsigned stream_grid(size_t items, int block) {
  size_t g = (items + block - 1) / block;
  const size_t cap = 256 * 8 * 4;
  return (unsigned)(g < cap ? (g ? g : 1) : cap);
}

// ---------------------------------------------------------------------------------------------------------------
// K1 (CKKS multiply front end)
// ---------------------------------------------------------------------------------------------------------------
template <int LB>
__global__ __launch_bounds__((1 << LB) / 16) void k_fused_tensor_intt(DevCtx c, const u64 *__restrict__ a,
                                                                      const u64 *__restrict__ b, u64 *__restrict__ c01,
                                                                      u64 *__restrict__ c2coef, u64 *__restrict__ c2ntt, int nl) {
  __shared__ u64 lds[lds_words(LB)];
  const int j = blockIdx.x % nl;
  const size_t ct = blockIdx.x / nl;
  const size_t N = (size_t)1 << LB, pw = (size_t)nl * N;
  const Mod m = c.mods[j];
  const NttTable t = ntt_table(c, j);
  const u64 *__restrict__ a0 = a + ct * 2 * pw + j * N, *__restrict__ a1 = a0 + pw;
  const u64 *__restrict__ b0 = b + ct * 2 * pw + j * N, *__restrict__ b1 = b0 + pw;
  u64 *__restrict__ o0 = c01 + ct * 2 * pw + j * N, *__restrict__ o1 = o0 + pw;
  u64 *__restrict__ dcoef = c2coef + (ct * nl + j) * N, *__restrict__ dntt = c2ntt + (ct * nl + j) * N;
  ntt_inv_block<LB>(
      lds,
      [&](int, int i) {
        const u64 x0 = a0[i], x1 = a1[i], y0 = b0[i], y1 = b1[i];
        o0[i] = mul_mod(x0, y0, m);
        U128 acc = mul_wide(x0, y1);
        mac128(acc, x1, y0);
        o1[i] = barrett_reduce(acc, m);
        const u64 v = mul_mod(x1, y1, m);
        dntt[i] = v;
        return v;
      },
      [&](int, int i, u64 v) { dcoef[i] = scale_inv_n(v, m); }, t, m, 0, 0);
}

// inverse transform of the key-switch operand when it arrives in NTT form (CKKS rotations):
// src limb (ct, j) at src + ct*src_stride + j*N  ->  dst[ct][j]
template <int LB>
__global__ __launch_bounds__((1 << LB) / 16) void k_fused_operand_intt(DevCtx c, const u64 *__restrict__ src, size_t src_stride,
                                                                       u64 *__restrict__ dst, int nl) {
  __shared__ u64 lds[lds_words(LB)];
  const size_t N = (size_t)1 << LB;
  const int j = blockIdx.x % nl;
  const size_t ct = blockIdx.x / nl;
  const Mod m = c.mods[j];
  const NttTable t = ntt_table(c, j);
  const u64 *__restrict__ s = src + ct * src_stride + (size_t)j * N;
  u64 *__restrict__ d = dst + (size_t)blockIdx.x * N;
  ntt_inv_block<LB>(
      lds, [&](int, int i) { return s[i]; }, [&](int, int i, u64 v) { d[i] = scale_inv_n(v, m); }, t, m, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------------
// key switch: operand in coefficient form at coef + ct*coef_stride + J*N; CKKS additionally has the operand's own
// NTT form at ntt + ct*ntt_stride + J*N (used where q_J is the key prime: switch_key_inplace, CKKS branch)
// ---------------------------------------------------------------------------------------------------------------
// K2a: dec[ct][I][J] = NTT_I( operand_J mod q_I )
// LAZY (unguarded transforms over primes <= 55 bits): the transform's raw outputs (< 64q) are stored as they are;
// the inner product accumulates 128-bit products and reduces once, so it needs no canonical operands.
template <int LB, bool GUARD, bool LAZY>
__global__ __launch_bounds__((1 << LB) / 16) void k_fused_ks_decomp_ntt(DevCtx c, const u64 *__restrict__ coef, size_t coef_stride,
                                                                        u64 *__restrict__ dec, int nl, int skip_diagonal) {
  __shared__ u64 lds[lds_words(LB)];
  const int J = blockIdx.x % nl;
  const int I = (blockIdx.x / nl) % (nl + 1);
  const size_t ct = blockIdx.x / ((size_t)nl * (nl + 1));
  if (skip_diagonal && J == I) return;
  const size_t N = (size_t)1 << LB;
  const int ki = (I == nl) ? c.K - 1 : I;
  const Mod m = c.mods[ki];
  const NttTable t = ntt_table(c, ki);
  const u64 *__restrict__ src = coef + ct * coef_stride + (size_t)J * N;
  u64 *__restrict__ dst = dec + ((ct * (nl + 1) + I) * nl + J) * N;
  // residues are < q_J already; the guarded transform accepts inputs < 4q, the unguarded one < 8q (57q of growth
  // on top stays below the 64q its consumers assume).  Workgroup-uniform choice between two straight-line bodies,
  // so the sixteen coefficient loads of a lane are issued back to back in either.
  const bool need_reduce = GUARD ? (c.mods[J].q > m.q) : ((c.mods[J].q >> 3) >= m.q);
  auto st = [&](int, int i, u64 v) { dst[i] = LAZY ? v : canon_fwd<GUARD>(v, m); };
  if (need_reduce)
    ntt_fwd_block<LB, GUARD>(lds, [&](int, int i) { return reduce64(src[i], m); }, st, t, m, 0, 0);
  else
    ntt_fwd_block<LB, GUARD>(lds, [&](int, int i) { return src[i]; }, st, t, m, 0, 0);
}

// K2b: acc_comp[k] = sum_J x_J[k] * key[J][comp][I][k]; data primes -> ksacc[ct][comp][I], special -> tsp[ct][comp]
__global__ __launch_bounds__(256) void k_fused_ks_mac(DevCtx c, const u64 *__restrict__ dec, const u64 *__restrict__ ntt,
                                                      size_t ntt_stride, const u64 *__restrict__ key, u64 *__restrict__ ksacc,
                                                      u64 *__restrict__ tsp, int nl, size_t count) {
  const size_t N = (size_t)c.n;
  const size_t per_ct = (size_t)(nl + 1) * (N / 2);
  const size_t items = count * per_ct;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const size_t ct = it / per_ct, r = it % per_ct;
    const int I = (int)(r / (N / 2));
    const size_t k = (r % (N / 2)) * 2;
    const int ki = (I == nl) ? c.K - 1 : I;
    const Mod m = c.mods[ki];
    U128 a00{0, 0}, a01{0, 0}, a10{0, 0}, a11{0, 0};
    for (int J = 0; J < nl; J++) {
      const u64 *xs = (ntt && J == I) ? ntt + ct * ntt_stride + (size_t)J * N + k : dec + ((ct * (nl + 1) + I) * nl + J) * N + k;
      const u64x2 x = *reinterpret_cast<const u64x2 *>(xs);
      const u64x2 k0 = *reinterpret_cast<const u64x2 *>(key + (((size_t)J * 2 + 0) * c.K + ki) * N + k);
      const u64x2 k1 = *reinterpret_cast<const u64x2 *>(key + (((size_t)J * 2 + 1) * c.K + ki) * N + k);
      mac128(a00, x.x, k0.x); mac128(a01, x.y, k0.y);
      mac128(a10, x.x, k1.x); mac128(a11, x.y, k1.y);
      if ((J & 3) == 3 || J == nl - 1) {  // flush: barrett_reduce admits 4 products of reduced operands
        a00 = U128{barrett_reduce(a00, m), 0}; a01 = U128{barrett_reduce(a01, m), 0};
        a10 = U128{barrett_reduce(a10, m), 0}; a11 = U128{barrett_reduce(a11, m), 0};
      }
    }
    u64x2 r0{a00.lo, a01.lo}, r1{a10.lo, a11.lo};
    if (I == nl) {
      *reinterpret_cast<u64x2 *>(tsp + (ct * 2 + 0) * N + k) = r0;
      *reinterpret_cast<u64x2 *>(tsp + (ct * 2 + 1) * N + k) = r1;
    } else {
      *reinterpret_cast<u64x2 *>(ksacc + ((ct * 2 + 0) * nl + I) * N + k) = r0;
      *reinterpret_cast<u64x2 *>(ksacc + ((ct * 2 + 1) * nl + I) * N + k) = r1;
    }
  }
}

// K2c: special-prime limb back to coefficients, plus q_sp/2 (rounding of the division by q_sp)
template <int LB>
__global__ __launch_bounds__((1 << LB) / 16) void k_fused_ks_special_intt(DevCtx c, const u64 *__restrict__ tsp,
                                                                          u64 *__restrict__ tlast) {
  __shared__ u64 lds[lds_words(LB)];
  const size_t N = (size_t)1 << LB;
  const Mod m = c.mods[c.K - 1];
  const NttTable t = ntt_table(c, c.K - 1);
  const u64 half = m.q >> 1;
  const u64 *__restrict__ src = tsp + (size_t)blockIdx.x * N;
  u64 *__restrict__ dst = tlast + (size_t)blockIdx.x * N;
  ntt_inv_block<LB>(
      lds, [&](int, int i) { return src[i]; }, [&](int, int i, u64 v) { dst[i] = add_mod(scale_inv_n(v, m), half, m.q); }, t, m,
      0, 0);
}

// K3, CKKS: out[ct][comp][j] = (ksacc - NTT_j(tlast mod q_j + fix)) * q_sp^-1 (+ addend)
template <int LB, bool GUARD>
__global__ __launch_bounds__((1 << LB) / 16) void k_fused_ks_moddown(DevCtx c, const u64 *__restrict__ ksacc,
                                                                     const u64 *__restrict__ tlast, const u64 *__restrict__ addend,
                                                                     size_t addend_stride, int add_c1, u64 *__restrict__ out, int nl) {
  __shared__ u64 lds[lds_words(LB)];
  const int j = blockIdx.x % nl;
  const size_t cc = blockIdx.x / nl;  // ct*2 + comp
  const size_t ct = cc >> 1;
  const int comp = (int)(cc & 1);
  const size_t N = (size_t)1 << LB;
  const Mod m = c.mods[j];
  const NttTable t = ntt_table(c, j);
  const u64 half = c.mods[c.K - 1].q >> 1;
  const u64 hm = reduce64(half, m);
  const u64 fix = hm ? m.q - hm : 0;
  const u64 inv = c.cst->inv_special[j], inv_s = c.cst->inv_special_s[j];
  const u64 *__restrict__ src = tlast + cc * N;
  const u64 *__restrict__ ks = ksacc + (cc * nl + j) * N;
  u64 *__restrict__ o = out + (cc * nl + j) * N;
  auto ld = [&](int, int i) { return add_mod(reduce64(src[i], m), fix, m.q); };
  if (addend && (comp == 0 || add_c1)) {  // workgroup-uniform: two straight-line bodies, no per-element select
    const u64 *__restrict__ cin = addend + ct * addend_stride + ((size_t)comp * nl + j) * N;
    ntt_fwd_block<LB, GUARD>(
        lds, ld,
        [&](int, int i, u64 v) {  // unguarded: v < 64q, so ks + 64q - v stays positive and mul_shoup reduces it
          const u64 d = GUARD ? sub_mod(ks[i], canon4(v, m), m.q) : ks[i] + (m.two_q << 5) - v;
          o[i] = add_mod(mul_shoup(d, inv, inv_s, m.q), cin[i], m.q);
        },
        t, m, 0, 0);
  } else {
    ntt_fwd_block<LB, GUARD>(
        lds, ld,
        [&](int, int i, u64 v) {
          const u64 d = GUARD ? sub_mod(ks[i], canon4(v, m), m.q) : ks[i] + (m.two_q << 5) - v;
          o[i] = mul_shoup(d, inv, inv_s, m.q);
        },
        t, m, 0, 0);
  }
}

// K3, BFV: out[ct][comp][j] = (INTT_j(ksacc) - (tlast mod q_j + fix)) * q_sp^-1 (+ addend), coefficient form
template <int LB>
__global__ __launch_bounds__((1 << LB) / 16) void k_fused_ks_moddown_bfv(DevCtx c, const u64 *__restrict__ ksacc,
                                                                         const u64 *__restrict__ tlast,
                                                                         const u64 *__restrict__ addend, size_t addend_stride,
                                                                         int add_c1, u64 *__restrict__ out, int nl) {
  __shared__ u64 lds[lds_words(LB)];
  const int j = blockIdx.x % nl;
  const size_t cc = blockIdx.x / nl;
  const size_t ct = cc >> 1;
  const int comp = (int)(cc & 1);
  const size_t N = (size_t)1 << LB;
  const Mod m = c.mods[j];
  const NttTable t = ntt_table(c, j);
  const u64 half = c.mods[c.K - 1].q >> 1;
  const u64 hm = reduce64(half, m);
  const u64 fix = hm ? m.q - hm : 0;
  const u64 inv = c.cst->inv_special[j], inv_s = c.cst->inv_special_s[j];
  const u64 *__restrict__ src = tlast + cc * N;
  const u64 *__restrict__ ks = ksacc + (cc * nl + j) * N;
  u64 *__restrict__ o = out + (cc * nl + j) * N;
  auto ld = [&](int, int i) { return ks[i]; };
  if (addend && (comp == 0 || add_c1)) {
    const u64 *__restrict__ cin = addend + ct * addend_stride + ((size_t)comp * nl + j) * N;
    ntt_inv_block<LB>(
        lds, ld,
        [&](int, int i, u64 v) {
          const u64 x = add_mod(reduce64(src[i], m), fix, m.q);
          o[i] = add_mod(mul_shoup(sub_mod(scale_inv_n(v, m), x, m.q), inv, inv_s, m.q), cin[i], m.q);
        },
        t, m, 0, 0);
  } else {
    ntt_inv_block<LB>(
        lds, ld,
        [&](int, int i, u64 v) {
          const u64 x = add_mod(reduce64(src[i], m), fix, m.q);
          o[i] = mul_shoup(sub_mod(scale_inv_n(v, m), x, m.q), inv, inv_s, m.q);
        },
        t, m, 0, 0);
  }
}

// ---- fp64 twins of key-switching keys -----------------------------------------------------------------------------------
// The split kernels multiply every key word into an fp64 residue: as u64 it costs a conversion per use (two instructions, sixteen
// words per thread of the last step); as a centred double, converted once when the key is first used, nothing.  Same layout
// [digit][2][K][N]; words modulo primes above 2^52 convert inexactly and are never read (the fp64 kernels touch fp64-capable
// primes only).
__global__ __launch_bounds__(256) void k_key_to_fp(DevCtx c, const u64 *__restrict__ key, double *__restrict__ keyf, size_t words) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += stride) {
    const int kp = (int)((i >> c.logn) % (size_t)c.K);
    const u64 q = c.mods[kp].q, v = key[i];
    keyf[i] = v > (q >> 1) ? -(double)(q - v) : (double)v;
  }
}
const double *key_twin_lookup(const abc_hip_ctx *c, const u64 *key) {  // no building: safe after the lanes have forked
  if (c->sw.no_key_twin) return nullptr;
  auto it = c->key_twins.find(key);
  return it == c->key_twins.end() ? nullptr : it->second;
}
// call BEFORE fork_lanes: the conversion runs on c->stream and the lanes wait for an event recorded behind it
const double *key_twin(abc_hip_ctx *c, const u64 *key) {
  if (c->sw.no_key_twin || !key) return nullptr;
  auto it = c->key_twins.find(key);
  if (it != c->key_twins.end()) return it->second;
  if (c->capture_active) return nullptr;  // built by the eager pass that precedes every recording
  double *d = nullptr;
  const size_t words = c->key_words();
  if (hipMalloc(&d, words * 8) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  hipLaunchKernelGGL(k_key_to_fp, dim3(stream_grid(words, 256)), dim3(256), 0, c->stream, c->dc, key, d, words);
  c->key_twins[key] = d;
  return d;
}
void drop_key_twins(abc_hip_ctx *c, const u64 *key) {
  if (c->key_twins.empty() && c->key_shoups.empty()) return;
  (void)hipStreamSynchronize(c->stream);
  auto drop = [&](auto &map) {
    if (!key) {
      for (auto &kv : map) (void)hipFree(kv.second);
      map.clear();
      return;
    }
    auto it = map.find(key);
    if (it != map.end()) {
      (void)hipFree(it->second);
      map.erase(it);
    }
  };
  drop(c->key_twins);
  drop(c->key_shoups);
}

// ---------------------------------------------------------------------------------------------------------------
// fp64 twins of the transform kernels above, taken when every key prime is below 2^50 (abc_ntt.hpp, "fp64 residue
// arithmetic").  Same buffers, same u64 memory format, bit-identical results; only the arithmetic between the HBM
// load and the HBM store differs.
// ---------------------------------------------------------------------------------------------------------------
// x*y mod q for two residues |x|, |y| <= q: |result| < q  (|x y / q| 2^-52 <= 1/4, plus the rounding 1/2)
__device__ __forceinline__ double fp_mulmod(double x, double y, double q, double qinv) {
  const double h = x * y;
  const double l = __builtin_fma(x, y, -h);
  const double c = __builtin_rint(h * qinv);
  return __builtin_fma(-c, q, h) + l;
}
// four consecutive words as two adjacent 16-byte stores
__device__ __forceinline__ void store_run4(u64 *p, const u64 (&v)[4]) {
  reinterpret_cast<u64x2 *>(p)[0] = u64x2{v[0], v[1]};
  reinterpret_cast<u64x2 *>(p)[1] = u64x2{v[2], v[3]};
}
// |r| < q -> canonical u64
__device__ __forceinline__ u64 fp_small_to_canon(double r, double q) {
  const u32 neg = (u32)((int)(u32)((u64)__double_as_longlong(r) >> 32) >> 31);
  const u64 qb = (u64)__double_as_longlong(q);
  const u64 add = ((u64)((u32)(qb >> 32) & neg) << 32) | (u64)((u32)qb & neg);
  r += __longlong_as_double((long long)add);
  return (u64)__double_as_longlong(r + 4503599627370496.0) & 0x000fffffffffffffull;
}

// CKKS tensor product of one limb, as the load functor of the inverse transform of c2: element i of (a0,a1) x (b0,b1)
// -> c0, c1 and c2 (NTT form) to memory in whole 32-byte runs, c2 returned for the transform.  A lane owns runs of
// four consecutive words; writing the halves of a run far apart reached HBM as two partial-sector writes (PMC
// WRITE_SIZE 35 instead of 28 limbs per multiply), hence the staging of four values and the back-to-back stores.
struct TensorFront {
  const u64 *__restrict__ a0, *__restrict__ a1, *__restrict__ b0, *__restrict__ b1;
  u64 *__restrict__ o0, *__restrict__ o1, *__restrict__ dntt;
  double q, qinv;
  u64 t0[4], t1[4], t2[4];
  __device__ __forceinline__ TensorFront(const u64 *a, const u64 *b, u64 *c01, u64 *c2ntt, size_t ct, int j, int nl, size_t N,
                                         const Mod &m)
      : q(m.qd), qinv(m.qinv) {
    const size_t pw = (size_t)nl * N;
    a0 = a + ct * 2 * pw + j * N; a1 = a0 + pw;
    b0 = b + ct * 2 * pw + j * N; b1 = b0 + pw;
    o0 = c01 + ct * 2 * pw + j * N; o1 = o0 + pw;
    dntt = c2ntt + (ct * nl + j) * N;
  }
  __device__ __forceinline__ double operator()(int r, int i) {
    const double x0 = fp_from_u64(a0[i]), x1 = fp_from_u64(a1[i]), y0 = fp_from_u64(b0[i]), y1 = fp_from_u64(b1[i]);
    const double v = fp_mulmod(x1, y1, q, qinv);
    t0[r & 3] = fp_small_to_canon(fp_mulmod(x0, y0, q, qinv), q);
    t1[r & 3] = fp_to_canon(fp_mulmod(x0, y1, q, qinv) + fp_mulmod(x1, y0, q, qinv), q, qinv);
    t2[r & 3] = fp_small_to_canon(v, q);
    if ((r & 3) == 3) {
      store_run4(o0 + i - 3, t0);
      store_run4(o1 + i - 3, t1);
      store_run4(dntt + i - 3, t2);
    }
    return v;
  }
};

template <int LB>
__global__ __launch_bounds__((1 << LB) / 16) void k_fused_operand_intt_fp(DevCtx c, const u64 *__restrict__ src, size_t src_stride,
                                                                          u64 *__restrict__ dst, int nl) {
  __shared__ double lds[lds_words(LB)];
  const size_t N = (size_t)1 << LB;
  const int j = blockIdx.x % nl;
  const size_t ct = blockIdx.x / nl;
  const Mod m = mod_at(c, j);
  const FpTable t = fp_table(c, j);
  const u64 *__restrict__ s = src + ct * src_stride + (size_t)j * N;
  u64 *__restrict__ d = dst + (size_t)blockIdx.x * N;
  ntt_inv_block_a<LB, FpArith>(
      lds, [&](int, int i) { return fp_from_u64(s[i]); },
      [&](int, int i, double v) { d[i] = fp_to_canon(fp_mul_lazy(v, m.inv_n_c, m.inv_n_cq, m.qd), m.qd, m.qinv); }, t, m, 0, 0);
}

// K2a.  No reduction of the operand limb modulo the target prime is needed at all: a forward transform tolerates
// inputs up to 2^50 for targets below 2^49 (growth (1 + 2^-5)^14), and the 49/50-bit targets, which re-centre at
// every pass, accept the at most 2q such an operand amounts to.  Outputs are stored as non-negative lazy
// representatives in [q/2, 3q/2] (the inner product reduces anyway).
template <int LB>
__global__ __launch_bounds__((1 << LB) / 16) void k_fused_ks_decomp_ntt_fp(DevCtx c, const u64 *__restrict__ coef, size_t coef_stride,
                                                                           u64 *__restrict__ dec, int nl, int skip_diagonal) {
  __shared__ double lds[lds_words(LB)];
  const int J = blockIdx.x % nl;
  const int I = (blockIdx.x / nl) % (nl + 1);
  const size_t ct = blockIdx.x / ((size_t)nl * (nl + 1));
  if (skip_diagonal && J == I) return;
  const size_t N = (size_t)1 << LB;
  const int ki = (I == nl) ? c.K - 1 : I;
  const Mod m = mod_at(c, ki);
  const FpTable t = fp_table(c, ki);
  const u64 *__restrict__ src = coef + ct * coef_stride + (size_t)J * N;
  u64 *__restrict__ dst = dec + ((ct * (nl + 1) + I) * nl + J) * N;
  ntt_fwd_block_a<LB, FpArith>(
      lds, [&](int, int i) { return fp_from_u64(src[i]); }, [&](int, int i, double v) { dst[i] = fp_to_lazy(v, m.qd, m.qinv); }, t,
      m, 0, 0);
}

template <int LB>
__global__ __launch_bounds__((1 << LB) / 16) void k_fused_ks_special_intt_fp(DevCtx c, const u64 *__restrict__ tsp,
                                                                             u64 *__restrict__ tlast) {
  __shared__ double lds[lds_words(LB)];
  const size_t N = (size_t)1 << LB;
  const Mod m = mod_at(c, c.K - 1);
  const FpTable t = fp_table(c, c.K - 1);
  const double half = (double)(m.q >> 1);
  const u64 *__restrict__ src = tsp + (size_t)blockIdx.x * N;
  u64 *__restrict__ dst = tlast + (size_t)blockIdx.x * N;
  ntt_inv_block_a<LB, FpArith>(
      lds, [&](int, int i) { return fp_from_u64(src[i]); },
      [&](int, int i, double v) { dst[i] = fp_to_canon(fp_mul_lazy(v, m.inv_n_c, m.inv_n_cq, m.qd) + half, m.qd, m.qinv); }, t, m, 0,
      0);
}

// K3, CKKS.  The special-prime polynomial (< q_sp <= 2^50) plus the rounding fix enters the transform unreduced
// (see K2a); the subtraction, the scaling by q_sp^-1 and the addend stay in doubles until the single
// canonicalisation of the store.
template <int LB, bool GAL>
__global__ __launch_bounds__((1 << LB) / 16) void k_fused_ks_moddown_fp(DevCtx c, const u64 *__restrict__ ksacc,
                                                                        const u64 *__restrict__ tlast, const u64 *__restrict__ addend,
                                                                        size_t addend_stride, int add_c1, u64 *__restrict__ out, int nl,
                                                                        int ncc, u32 gelt) {
  __shared__ double lds[lds_words(LB)];
  // The nl workgroups that read the same special-prime polynomial (ct, comp) are 8 apart in blockIdx: workgroups go
  // round-robin over the 8 XCDs, so they share one L2 and the polynomial is fetched from HBM once, not nl times.
  const unsigned per = 8u * (unsigned)nl;
  const unsigned grp = blockIdx.x / per, rem = blockIdx.x % per;
  const unsigned left = (unsigned)ncc - grp * 8u, gsz = left < 8u ? left : 8u;  // the last group may be ragged
  const int j = (int)(rem / gsz);
  const size_t cc = (size_t)grp * 8 + rem % gsz;  // ct*2 + comp
  const size_t ct = cc >> 1;
  const int comp = (int)(cc & 1);
  const size_t N = (size_t)1 << LB;
  const Mod m = mod_at(c, j);
  const FpTable t = fp_table(c, j);
  const u64 half = c.mods[c.K - 1].q >> 1;
  const u64 hm = reduce64(half, m);
  const double fix = hm ? (double)(m.q - hm) : 0.0;
  const double inv = c.cst->inv_special_c[j], inv_q = c.cst->inv_special_cq[j];
  const u64 *__restrict__ src = tlast + cc * N;
  const u64 *__restrict__ ks = ksacc + (cc * nl + j) * N;
  u64 *__restrict__ o = out + (cc * nl + j) * N;
  auto ld = [&](int, int i) { return fp_from_u64(src[i]) + fix; };
  if (addend && (comp == 0 || add_c1)) {  // workgroup-uniform: two straight-line bodies, no per-element select
    const u64 *__restrict__ cin = addend + ct * addend_stride + ((size_t)comp * nl + j) * N;
    ntt_fwd_block_a<LB, FpArith>(
        lds, ld,
        [&](int, int i, double v) {
          const double d = fp_from_u64(ks[i]) - v;
          o[i] = fp_to_canon(fp_mul_lazy(d, inv, inv_q, m.qd) + fp_from_u64(cin[galois_ntt_src<GAL>((u32)i, gelt, LB)]), m.qd, m.qinv);
        },
        t, m, 0, 0);
  } else {
    ntt_fwd_block_a<LB, FpArith>(
        lds, ld,
        [&](int, int i, double v) {
          const double d = fp_from_u64(ks[i]) - v;
          o[i] = fp_to_canon(fp_mul_lazy(d, inv, inv_q, m.qd), m.qd, m.qinv);
        },
        t, m, 0, 0);
  }
}

// K3, BFV
template <int LB>
__global__ __launch_bounds__((1 << LB) / 16) void k_fused_ks_moddown_bfv_fp(DevCtx c, const u64 *__restrict__ ksacc,
                                                                            const u64 *__restrict__ tlast,
                                                                            const u64 *__restrict__ addend, size_t addend_stride,
                                                                            int add_c1, u64 *__restrict__ out, int nl) {
  __shared__ double lds[lds_words(LB)];
  const int j = blockIdx.x % nl;
  const size_t cc = blockIdx.x / nl;
  const size_t ct = cc >> 1;
  const int comp = (int)(cc & 1);
  const size_t N = (size_t)1 << LB;
  const Mod m = mod_at(c, j);
  const FpTable t = fp_table(c, j);
  const u64 half = c.mods[c.K - 1].q >> 1;
  const u64 hm = reduce64(half, m);
  const double fix = hm ? (double)(m.q - hm) : 0.0;
  const double inv = c.cst->inv_special_c[j], inv_q = c.cst->inv_special_cq[j];
  const u64 *__restrict__ src = tlast + cc * N;
  const u64 *__restrict__ ks = ksacc + (cc * nl + j) * N;
  u64 *__restrict__ o = out + (cc * nl + j) * N;
  auto ld = [&](int, int i) { return fp_from_u64(ks[i]); };
  if (addend && (comp == 0 || add_c1)) {
    const u64 *__restrict__ cin = addend + ct * addend_stride + ((size_t)comp * nl + j) * N;
    ntt_inv_block_a<LB, FpArith>(
        lds, ld,
        [&](int, int i, double v) {
          const double d = fp_mul_lazy(v, m.inv_n_c, m.inv_n_cq, m.qd) - (fp_from_u64(src[i]) + fix);
          o[i] = fp_to_canon(fp_mul_lazy(d, inv, inv_q, m.qd) + fp_from_u64(cin[i]), m.qd, m.qinv);
        },
        t, m, 0, 0);
  } else {
    ntt_inv_block_a<LB, FpArith>(
        lds, ld,
        [&](int, int i, double v) {
          const double d = fp_mul_lazy(v, m.inv_n_c, m.inv_n_cq, m.qd) - (fp_from_u64(src[i]) + fix);
          o[i] = fp_to_canon(fp_mul_lazy(d, inv, inv_q, m.qd), m.qd, m.qinv);
        },
        t, m, 0, 0);
  }
}

// K1 + K2a in one workgroup (CKKS multiply): the coefficients of c2_j leave the inverse transform in exactly the
// register layout the forward transform's first pass loads (PassIdx<LB,0,R0> both ways), so they stay in 16
// registers per lane and feed the nl forward transforms modulo the other key primes back to back -- the
// coefficient form of c2 never touches HBM (saves 4 + 16 limb transfers per ciphertext and one launch).
template <int LB>
__global__ __launch_bounds__((1 << LB) / 16) void k_fused_tensor_decomp_fp(DevCtx c, const u64 *__restrict__ a,
                                                                           const u64 *__restrict__ b, u64 *__restrict__ c01,
                                                                           u64 *__restrict__ c2ntt, u64 *__restrict__ dec, int nl) {
  __shared__ double lds[lds_words(LB)];
  const int j = blockIdx.x % nl;
  const size_t ct = blockIdx.x / nl;
  const size_t N = (size_t)1 << LB;
  double src[16];
  {
    const Mod m = mod_at(c, j);
    const FpTable t = fp_table(c, j);
    TensorFront front(a, b, c01, c2ntt, ct, j, nl, N, m);
    ntt_inv_block_a<LB, FpArith>(
        lds, [&](int r, int i) { return front(r, i); },
        // canonical [0, q_j) as a double: the value SEAL's decomposition reduces modulo the other primes
        [&](int r, int, double v) {
          const double w = fp_centre(fp_mul_lazy(v, m.inv_n_c, m.inv_n_cq, m.qd), m.qd, m.qinv);
          src[r] = w < 0.0 ? w + m.qd : w;
        },
        t, m, 0, 0);
  }
  for (int I = 0; I <= nl; I++) {
    if (I == j) continue;
    const int ki = (I == nl) ? c.K - 1 : I;
    const Mod m = mod_at(c, ki);
    const FpTable t = fp_table(c, ki);
    u64 *__restrict__ dst = dec + ((ct * (nl + 1) + I) * nl + j) * N;
    block_sync_lds();  // the previous transform's last pass has read its LDS words
    ntt_fwd_block_a<LB, FpArith>(
        lds, [&](int r, int) { return src[r]; }, [&](int, int i, double v) { dst[i] = fp_to_lazy(v, m.qd, m.qinv); }, t, m, 0, 0);
  }
}

// the sixteen stride-2^(LB-4) values of a lane after the first register pass of a forward transform -> the half-done limb at dst,
// raw doubles (kind 0) or centred and packed (abc_ntt.hpp, "packed half-done limbs"); kind is workgroup-uniform
template <int LB>
__device__ __forceinline__ void store_half_done(double *__restrict__ dst, double (&y)[16], int p, const FpK &kk, int kind) {
  constexpr size_t N = (size_t)1 << LB;
  constexpr int SH = LB - 4;
  if (kind == 0) {
#pragma unroll
    for (int k = 0; k < 16; k++) dst[(k << SH) + p] = y[k];
  } else if (kind == 1) {
#pragma unroll
    for (int k = 0; k < 16; k++) pack_store<1>(dst, N, (size_t)(k << SH) + p, fp_centre(y[k], kk.q, kk.qinv));
  } else {
#pragma unroll
    for (int k = 0; k < 16; k++) pack_store<2>(dst, N, (size_t)(k << SH) + p, fp_centre(y[k], kk.q, kk.qinv));
  }
}

// ---- split transforms (N = 2^14) ------------------------------------------------------------------------------------
// A 2^14-point forward transform is a radix-16 pass over stride-1024 elements (stages 0..3, no LDS: the sixteen
// operands of a lane are exactly what the inverse transform's last pass leaves in its registers) followed by
// sixteen independent 1024-point transforms (stages 4..13) that one wavefront each completes in 8.5 KiB of LDS.
// The first kernel of a sequence does the LDS-resident part (tensor product or operand, inverse transform of limb j) and the
// register pass of the forward transforms, and stores the half-done limbs (raw doubles or packed, abc_ntt.hpp); the later
// kernels finish them block by block and multiply into the key on the fly.

// K1s for a general key switch.  CKKS (the operand arrives in NTT form): inverse transform of limb j in LDS, then the
// register pass modulo every other key prime.  BFV (coefficient form): no LDS at all, the register pass modulo every key
// prime including q_j itself.
template <int LB, bool CKKS, bool GAL>
__global__ __launch_bounds__((1 << LB) / 16) void k_fused_operand_pass0_fp(DevCtx c, const u64 *__restrict__ src, size_t src_stride,
                                                                           double *__restrict__ part, int nl, int per_target,
                                                                           u32 gelt, int padded) {
  static_assert(LB == 14, "split transforms are laid out for N = 2^14");
  __shared__ double lds[CKKS ? lds_words(LB) : 1];
  const size_t N = (size_t)1 << LB;
  // per_target (BFV, few ciphertexts in flight): one workgroup per (ct, J, target I) instead of per (ct, J), so a
  // single-ciphertext call spreads over nl (nl+1) CUs rather than nl
  const unsigned wg = per_target ? blockIdx.x / (unsigned)(nl + 1) : blockIdx.x;
  const int only = per_target ? (int)(blockIdx.x % (unsigned)(nl + 1)) : -1;
  const int j = wg % nl;
  const size_t ct = wg / nl;
  const int tid = threadIdx.x;
  const u64 *__restrict__ sp = src + ct * src_stride + (size_t)j * N;
  double x[16];
  if constexpr (CKKS) {
    const Mod m = mod_at(c, j);
    const FpTable t = fp_table(c, j);
    ntt_inv_block_a<LB, FpArith>(
        lds, [&](int, int i) { return fp_from_u64(sp[galois_ntt_src<GAL>((u32)i, gelt, LB)]); },  // gelt: rotation folded in
        [&](int r, int, double v) {
          double w = fp_centre(fp_mul_lazy(v, m.inv_n_c, m.inv_n_cq, m.qd), m.qd, m.qinv);
          x[r] = w < 0.0 ? w + m.qd : w;
        },
        t, m, 0, 0);
  } else if (gelt) {  // BFV rotation: gelt = elt^-1 mod 2N, the permutation (with its sign) folded into the load -- workgroup-uniform
    const u64 qj = c.mods[j].q;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      bool neg;
      const u64 v = sp[galois_coef_src((u32)((k << 10) + tid), gelt, LB, neg)];
      x[k] = fp_from_u64(neg ? neg_mod(v, qj) : v);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = fp_from_u64(sp[(k << 10) + tid]);
  }
  const int hi0[1] = {0};
  for (int I = 0; I <= nl; I++) {
    if (CKKS && I == j) continue;
    if (only >= 0 && I != only) continue;
    const int ki = (I == nl) ? c.K - 1 : I;
    const Mod m = mod_at(c, ki);
    const FpTable t = fp_table(c, ki);
    const FpK kk = FpArith::consts(m);
    double y[16];
#pragma unroll
    for (int k = 0; k < 16; k++) y[k] = x[k];
    fwd_pass<FpArith, LB, 0, 4>(y, hi0, t, kk, 0, 0);
    double *__restrict__ dst = part + ((ct * (nl + 1) + I) * nl + j) * ((padded & 1) ? (size_t)c.ps : N);
    store_half_done<LB>(dst, y, tid, kk, ((padded & 2) && I < nl) ? pack_kind(m.bits) : 0);  // padded bit 1: packed (data primes only)
  }
}

// ---- the split sequence of the hot call (N = 2^14, CKKS, every key prime < 2^50): four launches, 84 limb transfers -----------
//   K1  k_split2_tensor_pass0_fp (multiply) / k_fused_operand_pass0_fp (rotation, relinearise): reads only a1, b1 (c2 = a1 b1 is
//       all the key switch needs), inverse transform in LDS, register pass modulo every other key prime -> half-done limbs
//   K2a k_split_special_fp: the special prime's inner product and the block-local part of its inverse transform
//   K2b k_split3_pass_fp (registers only): last radix-16 pass of that inverse transform, N^-1, + q_sp/2, then for every data prime
//       q_j the first radix-16 pass of the forward transform of (t mod q_j): half-done mod-down limbs
//   K2c k_split4_main_fp (up to five data limbs) / k_split3_main_fp, per (ct, data prime I, block): nl - 1 wavefronts finish the
//       decomposition limbs, two more the mod-down limbs of K2b on the same block, then every thread forms, for two adjacent
//       coefficients,   out_c = (sum_J x_J key_J,c + q_sp c_c - NTT_I(t_c)) q_sp^-1   (mod q_I)
//       with the diagonal operand c2_I = a1 b1 and c0 = a0 b0, c1 = a0 b1 + a1 b0 computed right there from a and b and folded in
//       as q_sp c:  (acc + q_sp c - NTT(t)) q_sp^-1 = (acc - NTT(t)) q_sp^-1 + c.  Neither c0, c1 nor c2's NTT form ever travel
//       through scratch, the accumulators never leave the CU, sums over J are formed in registers (no LDS atomics), and `out`
//       may alias an operand (a and b are last read here, by the thread that then writes those words).
// (Earlier generations -- LDS-atomic tail kernels, the LDS-resident mod-down, an LDS-table twin of K2a -- were measured in
// rounds 1 and 2 and are gone: DESIGN.md section 4.)
template <int LB>
__global__ __launch_bounds__((1 << LB) / 16) void k_split2_tensor_pass0_fp(DevCtx c, const u64 *__restrict__ a,
                                                                           const u64 *__restrict__ b, double *__restrict__ part,
                                                                           int nl, int pack) {
  static_assert(LB == 14, "split transforms are laid out for N = 2^14");
  __shared__ double lds[lds_words(LB)];
  const int j = blockIdx.x % nl;
  const size_t ct = blockIdx.x / nl;
  const size_t N = (size_t)1 << LB, pw = (size_t)nl * N;
  double src[16];
  {
    const Mod m = mod_at(c, j);
    const FpTable t = fp_table(c, j);
    const u64 *__restrict__ a1 = a + ct * 2 * pw + pw + (size_t)j * N, *__restrict__ b1 = b + ct * 2 * pw + pw + (size_t)j * N;
    const double q = m.qd, qinv = m.qinv;
    ntt_inv_block_a<LB, FpArith>(
        lds, [&](int, int i) { return fp_mulmod(fp_from_u64(a1[i]), fp_from_u64(b1[i]), q, qinv); },
        // canonical [0, q_j) as a double: the value SEAL's decomposition reduces modulo the other primes
        [&](int r, int, double v) {
          const double w = fp_centre(fp_mul_lazy(v, m.inv_n_c, m.inv_n_cq, m.qd), m.qd, m.qinv);
          src[r] = w < 0.0 ? w + m.qd : w;
        },
        t, m, 0, 0);
  }
  const int tid = threadIdx.x;
  const int hi0[1] = {0};
  for (int I = 0; I <= nl; I++) {
    if (I == j) continue;
    const int ki = (I == nl) ? c.K - 1 : I;
    const Mod m = mod_at(c, ki);
    const FpTable t = fp_table(c, ki);
    const FpK kk = FpArith::consts(m);
    double y[16];
#pragma unroll
    for (int k = 0; k < 16; k++) y[k] = src[k];
    fwd_pass<FpArith, LB, 0, 4>(y, hi0, t, kk, 0, 0);
    double *__restrict__ dst = part + ((ct * (nl + 1) + I) * nl + j) * (size_t)c.ps;
    store_half_done<LB>(dst, y, tid, kk, (pack && I < nl) ? pack_kind(m.bits) : 0);  // the special prime's limbs stay raw
  }
}

// K2a, the special prime's share of the key inner product (grid (ct, block); nl wavefronts): wavefront J finishes stages 4..13 of
// the half-done limb (ct, special prime, J) on this 1024-point block and parks it in LDS; after one barrier every thread sums, for
// two adjacent coefficients, x_J key[J][c][special] over J in registers; the two sums go back to LDS and wavefronts 0 and 1 run
// the ten block-local stages of the special-prime limb's INVERSE transform on them -> tsp_half (raw doubles); k_split3_pass_fp
// does the remaining radix-16 pass.  NL > 0: compile-time limb count (the loop over J unrolls, all its loads are issued
// together); NL = 0: any nl <= 12.
template <int NL>
__global__ __launch_bounds__(NL ? NL * 64 : 768) void k_split_special_fp(DevCtx c, const double *__restrict__ part, const u64 *__restrict__ key,
                                                                         const double *__restrict__ keyf, double *__restrict__ tsp_half,
                                                                         int nl_rt) {
  extern __shared__ double dyn[];  // max(nl, 2) buffers of one 1024-point block each
  const int nl = NL ? NL : nl_rt;
  const int J = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int blk = blockIdx.x & 15;
  const size_t ct = (size_t)(blockIdx.x >> 4);
  const size_t N = (size_t)c.n, base = (size_t)blk << 10;
  const int ki = c.K - 1;
  const Mod m = mod_at(c, ki);
  const FpTable t = fp_table(c, ki);
  const double q = m.qd, qinv = m.qinv;
  {
    double *buf = dyn + J * lds_words(10);
    const double *__restrict__ src = part + ((ct * (nl + 1) + nl) * nl + J) * (size_t)c.ps + base;
    ntt_fwd_block_a<10, FpTail>(
        buf, [&](int, int i) { return fp_centre(src[i], q, qinv); }, [&](int, int i, double v) { buf[lds_pad(i)] = v; }, t, m, 4, blk,
        lane);
  }
  __syncthreads();
  for (int e = 2 * (int)threadIdx.x; e < 1024; e += 2 * (int)blockDim.x) {  // this thread: coefficients e, e + 1 of the block
    double s0[2] = {0.0, 0.0}, s1[2] = {0.0, 0.0};
#pragma unroll
    for (int Jx = 0; Jx < (NL ? NL : 12); Jx++) {
      if (!NL && Jx >= nl) break;
      const f64x2 v = *reinterpret_cast<const f64x2 *>(dyn + Jx * lds_words(10) + lds_pad(e));
      const size_t i0 = (((size_t)Jx * 2 + 0) * c.K + ki) * N + base + e, i1 = i0 + (size_t)c.K * N;
      f64x2 k0, k1;
      if (keyf) {  // the key's fp64 twin (workgroup-uniform): no conversion
        k0 = *reinterpret_cast<const f64x2 *>(keyf + i0);
        k1 = *reinterpret_cast<const f64x2 *>(keyf + i1);
      } else {
        const u64x2 r0 = *reinterpret_cast<const u64x2 *>(key + i0), r1 = *reinterpret_cast<const u64x2 *>(key + i1);
        k0.x = fp_from_u64(r0.x); k0.y = fp_from_u64(r0.y);
        k1.x = fp_from_u64(r1.x); k1.y = fp_from_u64(r1.y);
      }
      s0[0] += fp_mulmod(v.x, k0.x, q, qinv);
      s0[1] += fp_mulmod(v.y, k0.y, q, qinv);
      s1[0] += fp_mulmod(v.x, k1.x, q, qinv);
      s1[1] += fp_mulmod(v.y, k1.y, q, qinv);
      if (Jx == 7) {  // eight products of magnitude < q stay below 2^53; re-centre before adding more
#pragma unroll
        for (int k = 0; k < 2; k++) {
          s0[k] = fp_centre(s0[k], q, qinv);
          s1[k] = fp_centre(s1[k], q, qinv);
        }
      }
    }
    // park the two sums (centred) in buffers 0 and 1; this thread has read its words of them already
    f64x2 r;
    r.x = fp_centre(s0[0], q, qinv); r.y = fp_centre(s0[1], q, qinv);
    *reinterpret_cast<f64x2 *>(dyn + lds_pad(e)) = r;
    r.x = fp_centre(s1[0], q, qinv); r.y = fp_centre(s1[1], q, qinv);
    *reinterpret_cast<f64x2 *>(dyn + lds_words(10) + lds_pad(e)) = r;
  }
  __syncthreads();
  for (int comp = J; comp < 2; comp += nl) {  // one wavefront per component
    double *buf = dyn + comp * lds_words(10);
    double *__restrict__ dst = tsp_half + (ct * 2 + comp) * (size_t)c.ps + base;
    ntt_inv_block_a<10, FpArith>(
        buf, [&](int, int i) { return buf[lds_pad(i)]; }, [&](int, int i, double v) { dst[i] = v; }, t, m, 4, blk, lane);
  }
}

static void launch_split_special(hipStream_t st, abc_hip_ctx *c, size_t cc, int nl, const double *part, const u64 *key, double *tsp_half) {
  const dim3 grid((unsigned)(cc * 16)), block(64 * nl);
  const size_t lds = (size_t)((nl < 2 ? 2 : nl) * lds_words(10)) * 8;
  const double *keyf = key_twin_lookup(c, key);
#define ABC_SP(NLV) hipLaunchKernelGGL((k_split_special_fp<NLV>), grid, block, lds, st, c->dc, part, key, keyf, tsp_half, nl)
  switch (nl) {
    case 1: ABC_SP(1); break;
    case 2: ABC_SP(2); break;
    case 3: ABC_SP(3); break;
    case 4: ABC_SP(4); break;
    default: ABC_SP(0); break;
  }
#undef ABC_SP
}

// scratch limbs per ciphertext: coef L, ntt L, dec L(L+1), ksacc 2L, tsp 2, tlast 2, c01 2L
static inline size_t fused_scratch_limbs(int nl) { return (size_t)nl * (nl + 1) + 6 * (size_t)nl + 4; }

struct FusedScratch {
  u64 *coef, *ntt, *dec, *ksacc, *tsp, *tlast, *c01;
};
static inline FusedScratch carve(u64 *base, size_t chunk, int nl, size_t N) {
  FusedScratch s;
  s.coef = base;
  s.ntt = s.coef + chunk * nl * N;
  s.dec = s.ntt + chunk * nl * N;
  s.ksacc = s.dec + chunk * nl * (nl + 1) * N;
  s.tsp = s.ksacc + chunk * 2 * nl * N;
  s.tlast = s.tsp + chunk * 2 * N;
  s.c01 = s.tlast + chunk * 2 * N;
  return s;
}


template <int LB>
__global__ __launch_bounds__(256) void k_split3_pass_fp(DevCtx c, const double *__restrict__ tsp_half, double *__restrict__ tpart, int nl,
                                                        int pack) {
  static_assert(LB == 14, "split transforms are laid out for N = 2^14");
  const size_t cc = blockIdx.x >> 2;                                  // ct*2 + comp
  const int p = (int)((blockIdx.x & 3) << 8) + (int)threadIdx.x;      // position inside every 1024-point block
  const int hi0[1] = {0};
  double x[16];
  {
    const Mod ms = mod_at(c, c.K - 1);
    const FpTable ts = fp_table(c, c.K - 1);
    const FpK ks = FpArith::consts(ms);
    const double *__restrict__ src = tsp_half + cc * (size_t)c.ps;
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = src[(k << 10) + p];
    FpArith::centre16(x, ks);
    inv_pass<FpArith, LB, 0, 4>(x, hi0, ts, ks, 0, 0);
    const double half = (double)(ms.q >> 1);
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const double w = fp_centre(fp_mul_lazy(x[k], ms.inv_n_c, ms.inv_n_cq, ms.qd) + half, ms.qd, ms.qinv);
      x[k] = w < 0.0 ? w + ms.qd : w;  // canonical [0, q_sp): what SEAL reduces modulo q_j
    }
  }
  const u64 half = c.mods[c.K - 1].q >> 1;
  for (int j = 0; j < nl; j++) {
    const Mod m = mod_at(c, j);
    const FpTable t = fp_table(c, j);
    const FpK kk = FpArith::consts(m);
    const u64 hm = reduce64(half, m);
    const double fix = hm ? (double)(m.q - hm) : 0.0;
    double y[16];
#pragma unroll
    for (int k = 0; k < 16; k++) y[k] = x[k] + fix;
    fwd_pass<FpArith, LB, 0, 4>(y, hi0, t, kk, 0, 0);
    double *__restrict__ dst = tpart + (cc * nl + j) * (size_t)c.ps;
    store_half_done<LB>(dst, y, p, kk, pack ? pack_kind(m.bits) : 0);
  }
}

template <int MODE, bool GAL, int NL>
__global__ __launch_bounds__(NL ? (NL + 1) * 64 : 832) void k_split3_main_fp(DevCtx c, const double *__restrict__ part,
                                                                              const double *__restrict__ tpart,
                                                                              const u64 *__restrict__ opa, const u64 *__restrict__ opb,
                                                                              size_t opa_stride, size_t opb_stride, int add_c1,
                                                                              const u64 *__restrict__ key, u64 *__restrict__ out, int nl_rt,
                                                                              u32 gelt) {
  extern __shared__ double dyn[];  // nl + 1 buffers of one 1024-point block: nl - 1 decomposition limbs, 2 mod-down limbs
  const int nl = NL ? NL : nl_rt;
  const int W = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int blk = blockIdx.x & 15;
  const int I = (int)((blockIdx.x >> 4) % nl);
  const size_t ct = (size_t)((blockIdx.x >> 4) / nl);
  const size_t N = (size_t)c.n, base = (size_t)blk << 10;
  const Mod m = mod_at(c, I);
  const FpTable t = fp_table(c, I);
  const double q = m.qd, qinv = m.qinv;
  {
    double *buf = dyn + W * lds_words(10);
    const double *__restrict__ src;
    if (W < nl - 1) {
      const int J = W < I ? W : W + 1;
      src = part + ((ct * (nl + 1) + I) * nl + J) * (size_t)c.ps + base;
    } else {
      src = tpart + ((ct * 2 + (W - (nl - 1))) * nl + I) * (size_t)c.ps + base;
    }
    ntt_fwd_block_a<10, FpArith>(
        buf, [&](int, int i) { return fp_centre(src[i], q, qinv); }, [&](int, int i, double v) { buf[lds_pad(i)] = v; }, t, m, 4, blk,
        lane);
  }
  __syncthreads();
  const double inv = c.cst->inv_special_c[I], inv_q = c.cst->inv_special_cq[I];
  const size_t pw = (size_t)nl * N;
  const double *tt0 = dyn + (nl - 1) * lds_words(10), *tt1 = dyn + nl * lds_words(10);
  for (int e = 2 * (int)threadIdx.x; e < 1024; e += 2 * (int)blockDim.x) {  // this thread: coefficients e, e + 1 of the block
    double s0[2] = {0.0, 0.0}, s1[2] = {0.0, 0.0}, d0[2] = {0.0, 0.0}, d1[2] = {0.0, 0.0};
#pragma unroll
    for (int Jx = 0; Jx < (NL ? NL : 12); Jx++) {
      if (!NL && Jx >= nl) break;
      double x[2];
      if (Jx == I) {
        if (MODE == 0) {
          const u64 *pa = opa + ct * 2 * pw + (size_t)I * N + base + e, *pb = opb + ct * 2 * pw + (size_t)I * N + base + e;
          const u64x2 a0 = *reinterpret_cast<const u64x2 *>(pa), a1 = *reinterpret_cast<const u64x2 *>(pa + pw);
          const u64x2 b0 = *reinterpret_cast<const u64x2 *>(pb), b1 = *reinterpret_cast<const u64x2 *>(pb + pw);
          const double x0[2] = {fp_from_u64(a0.x), fp_from_u64(a0.y)}, x1[2] = {fp_from_u64(a1.x), fp_from_u64(a1.y)};
          const double y0[2] = {fp_from_u64(b0.x), fp_from_u64(b0.y)}, y1[2] = {fp_from_u64(b1.x), fp_from_u64(b1.y)};
#pragma unroll
          for (int k = 0; k < 2; k++) {
            x[k] = fp_mulmod(x1[k], y1[k], q, qinv);
            d0[k] = fp_mulmod(x0[k], y0[k], q, qinv);
            d1[k] = fp_mulmod(x0[k], y1[k], q, qinv) + fp_mulmod(x1[k], y0[k], q, qinv);
          }
        } else {
          const u64 *xl = opa + ct * opa_stride + (size_t)I * N;  // whole limb: a rotation gathers across blocks
#pragma unroll
          for (int k = 0; k < 2; k++) {
            const u32 si = galois_ntt_src<GAL>((u32)(base + e + k), gelt, c.logn);
            x[k] = fp_from_u64(xl[si]);
            if (opb) {
              const u64 *ad = opb + ct * opb_stride + (size_t)I * N;
              d0[k] = fp_from_u64(ad[si]);
              if (add_c1) d1[k] = fp_from_u64(ad[pw + si]);
            }
          }
        }
      } else {
        const int w = Jx < I ? Jx : Jx - 1;
        const f64x2 v = *reinterpret_cast<const f64x2 *>(dyn + w * lds_words(10) + lds_pad(e));
        x[0] = v.x;
        x[1] = v.y;
      }
      const u64x2 k0 = *reinterpret_cast<const u64x2 *>(key + (((size_t)Jx * 2 + 0) * c.K + I) * N + base + e);
      const u64x2 k1 = *reinterpret_cast<const u64x2 *>(key + (((size_t)Jx * 2 + 1) * c.K + I) * N + base + e);
      s0[0] += fp_mulmod(x[0], fp_from_u64(k0.x), q, qinv);
      s0[1] += fp_mulmod(x[1], fp_from_u64(k0.y), q, qinv);
      s1[0] += fp_mulmod(x[0], fp_from_u64(k1.x), q, qinv);
      s1[1] += fp_mulmod(x[1], fp_from_u64(k1.y), q, qinv);
      if (Jx == 7) {  // eight products of magnitude < q stay below 2^53; re-centre before adding more
#pragma unroll
        for (int k = 0; k < 2; k++) {
          s0[k] = fp_centre(s0[k], q, qinv);
          s1[k] = fp_centre(s1[k], q, qinv);
        }
      }
    }
    const f64x2 u0 = *reinterpret_cast<const f64x2 *>(tt0 + lds_pad(e)), u1 = *reinterpret_cast<const f64x2 *>(tt1 + lds_pad(e));
    // (sum + q_sp (c0, c1) - NTT(t)) q_sp^-1
    u64x2 r;
    r.x = fp_to_canon(fp_mul_lazy(s0[0] - u0.x, inv, inv_q, q) + d0[0], q, qinv);
    r.y = fp_to_canon(fp_mul_lazy(s0[1] - u0.y, inv, inv_q, q) + d0[1], q, qinv);
    *reinterpret_cast<u64x2 *>(out + ((ct * 2 + 0) * nl + I) * N + base + e) = r;
    r.x = fp_to_canon(fp_mul_lazy(s1[0] - u1.x, inv, inv_q, q) + d1[0], q, qinv);
    r.y = fp_to_canon(fp_mul_lazy(s1[1] - u1.y, inv, inv_q, q) + d1[1], q, qinv);
    *reinterpret_cast<u64x2 *>(out + ((ct * 2 + 1) * nl + I) * N + base + e) = r;
  }
}

template <int MODE, bool GAL>
static void launch_split3_main(hipStream_t st, abc_hip_ctx *c, size_t cc, int nl, const double *part, const double *tpart, const u64 *opa,
                               const u64 *opb, size_t opa_stride, size_t opb_stride, int add_c1, const u64 *key, u64 *out, u32 gelt) {
  const dim3 grid((unsigned)(cc * nl * 16)), block(64 * (nl + 1));
  const size_t lds = (size_t)((nl + 1) * lds_words(10)) * 8;
#define ABC_TM3(NLV)                                                                                                                 \
  hipLaunchKernelGGL((k_split3_main_fp<MODE, GAL, NLV>), grid, block, lds, st, c->dc, part, tpart, opa, opb, opa_stride, opb_stride, \
                     add_c1, key, out, nl, gelt)
  switch (nl) {
    case 1: ABC_TM3(1); break;
    case 2: ABC_TM3(2); break;
    case 3: ABC_TM3(3); break;
    case 4: ABC_TM3(4); break;
    default: ABC_TM3(0); break;
  }
#undef ABC_TM3
}


// ---- "split4": K2c with every vector-memory request issued up front ----------------------------------------------------------
// Same arithmetic and the same buffers as k_split3_main_fp.  What changes is the order of memory requests inside a wavefront:
// the per-lane twiddles of the tail transform come from an LDS table of the block's own twiddles (16 KiB per workgroup, filled
// cooperatively) instead of vector loads, so nothing in the transform touches the in-order vector-memory counter any more and
// the operands of the phase AFTER the transform (key slices, a and b) can be requested before it: one exposed memory latency per
// workgroup instead of two, and the transform runs under the second one.
template <int MODE, int NL>
struct PairOps {
  u64x2 k0[NL], k1[NL];
  u64x2 a0, a1, b0, b1;      // MODE 0
  u64 xs[2], d0s[2], d1s[2];  // MODE 1
};

template <int MODE, bool GAL, int NL>
__global__ __launch_bounds__(512, NL <= 4 ? 4 : 2) void k_split4_main_fp(DevCtx c, const double *__restrict__ part,
                                                                   const double *__restrict__ tpart, const u64 *__restrict__ opa,
                                                                   const u64 *__restrict__ opb, size_t opa_stride, size_t opb_stride,
                                                                   int add_c1, const u64 *__restrict__ key, u64 *__restrict__ out, u32 gelt,
                                                                   u32 imap, int ni, int pack, const double *__restrict__ keyf) {
  // grid (ct, slot, block), slot < ni; the data prime of a slot is nibble `slot` of imap (all of them: 0x76543210, ni = nl; a subset
  // when a chain mixes fp64-capable and wider primes: abc_kernels_isplit.hip)
  // pack: the half-done limbs of `part` / `tpart` modulo primes of at most 48 bits arrive packed (abc_ntt.hpp)
  extern __shared__ double dyn[];  // nl + 1 transform buffers, then the block's twiddle table (1024 {w, w/q} pairs)
  // 512 threads whatever nl: one coefficient pair per thread afterwards, so every operand of that phase is requested up front;
  // wavefronts nl + 1 .. 7 have no limb to transform and only take part in the table fill and the inner product
  static_assert(NL + 1 <= 8, "one wavefront per limb, eight wavefronts");
  constexpr int nl = NL, NT = 512, PER = 2;
  const int W = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int blk = blockIdx.x & 15;
  const int I = (int)((imap >> (4 * ((blockIdx.x >> 4) % (unsigned)ni))) & 15u);
  const size_t ct = (size_t)((blockIdx.x >> 4) / (unsigned)ni);
  const size_t N = (size_t)c.n, base = (size_t)blk << 10;
  const Mod m = mod_at(c, I);
  const FpTable t = fp_table(c, I);
  const double q = m.qd, qinv = m.qinv;
  f64x2 *ltw = reinterpret_cast<f64x2 *>(dyn + (nl + 1) * lds_words(10));
  const size_t pw = (size_t)nl * N;
  // scalar loads (constant address space): a vector load of these after the transform would expose one more memory latency
  const ABC_CONST_AS DevConst *cst = (const ABC_CONST_AS DevConst *)c.cst;
  const double inv = cst->inv_special_c[I], inv_q = cst->inv_special_cq[I];

  // (1) twiddle table, (2) this wavefront's half-done limb, (3) the operands of this thread's first coefficient pair
  f64x2 twv[PER];
  block_twiddles_fetch<10, f64x2, PER>(t.tw, 4, blk, (int)threadIdx.x, NT, twv);
  const bool has_limb = W <= nl;  // wavefront-uniform
  const int Wc = has_limb ? W : 0;
  const double *__restrict__ src = (Wc < nl - 1) ? part + ((ct * (nl + 1) + I) * nl + (Wc < I ? Wc : Wc + 1)) * (size_t)c.ps + base
                                                 : tpart + ((ct * 2 + (Wc - (nl - 1))) * nl + I) * (size_t)c.ps + base;
  const int pk = pack ? pack_kind(m.bits) : 0;  // workgroup-uniform
  u64x2 raw[8];
  if (has_limb) {  // eight loads of one coefficient pair: slot g*8 + k = element k*128 + 2*lane + g (ntt_fwd_tail1024_pairs)
    if (pk == 0) {
#pragma unroll
      for (int k = 0; k < 8; k++) raw[k] = *reinterpret_cast<const u64x2 *>(src + (k << 7) + 2 * lane);
    } else if (pk == 1) {
#pragma unroll
      for (int k = 0; k < 8; k++) raw[k] = pack_load_pair<1>(src - base, N, base + (k << 7) + 2 * lane);
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++) raw[k] = pack_load_pair<2>(src - base, N, base + (k << 7) + 2 * lane);
    }
  }
  auto load_pair = [&](int e, PairOps<MODE, NL> &o) {
    // key words: 16 raw bytes per (digit, component) either way -- the key's fp64 twin where it exists (keyf, workgroup-uniform:
    // the words ARE the doubles), the u64 key otherwise (converted when used, after the transform)
    const u64 *kw = keyf ? reinterpret_cast<const u64 *>(keyf) : key;
#pragma unroll
    for (int Jx = 0; Jx < NL; Jx++) {
      o.k0[Jx] = *reinterpret_cast<const u64x2 *>(kw + (((size_t)Jx * 2 + 0) * c.K + I) * N + base + e);
      o.k1[Jx] = *reinterpret_cast<const u64x2 *>(kw + (((size_t)Jx * 2 + 1) * c.K + I) * N + base + e);
    }
    if (MODE == 0) {
      const u64 *pa = opa + ct * 2 * pw + (size_t)I * N + base + e, *pb = opb + ct * 2 * pw + (size_t)I * N + base + e;
      o.a0 = *reinterpret_cast<const u64x2 *>(pa); o.a1 = *reinterpret_cast<const u64x2 *>(pa + pw);
      o.b0 = *reinterpret_cast<const u64x2 *>(pb); o.b1 = *reinterpret_cast<const u64x2 *>(pb + pw);
    } else {
      const u64 *xl = opa + ct * opa_stride + (size_t)I * N;  // whole limb: a rotation gathers across blocks
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
  };
  PairOps<MODE, NL> ops;
  load_pair(2 * (int)threadIdx.x, ops);

  block_twiddles_store<10, f64x2, PER>(ltw, (int)threadIdx.x, NT, twv);
  __syncthreads();
  if (has_limb) {
    double *buf = dyn + W * lds_words(10);
    double xin[16];
    if (pk == 0) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        xin[k] = __longlong_as_double((long long)raw[k].x);
        xin[8 + k] = __longlong_as_double((long long)raw[k].y);
      }
    } else if (pk == 1) {
#pragma unroll
      for (int k = 0; k < 8; k++) pack_decode_pair<1>(raw[k], xin[k], xin[8 + k]);
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++) pack_decode_pair<2>(raw[k], xin[k], xin[8 + k]);
    }
    // raw half-done limbs arrive below 4.1 q (four lazy stages from a canonical value): primes of 49 / 50 bits re-centre before
    // the remaining ten stages, smaller ones have the headroom for all fourteen (abc_ntt.hpp, FpK::red); packed ones arrive
    // centred -- wavefront-uniform branch
    if (m.bits >= 49) {
#pragma unroll
      for (int r = 0; r < 16; r++) xin[r] = fp_centre(xin[r], q, qinv);
    }
    ntt_fwd_tail1024_pairs<FpTail>(buf, xin, [&](int, int i, double v) { buf[lds_pad(i)] = v; }, t, m, 4, blk, lane, ltw);
  }
  __syncthreads();
  const double *tt0 = dyn + (nl - 1) * lds_words(10), *tt1 = dyn + nl * lds_words(10);
  auto compute_pair = [&](int e, const PairOps<MODE, NL> &o, auto twin) {
    constexpr bool TW = decltype(twin)::value;
    auto kd = [](u64 w) { return TW ? __longlong_as_double((long long)w) : fp_from_u64(w); };
    double s0[2] = {0.0, 0.0}, s1[2] = {0.0, 0.0}, d0[2] = {0.0, 0.0}, d1[2] = {0.0, 0.0};
#pragma unroll
    for (int Jx = 0; Jx < NL; Jx++) {
      double x[2];
      if (Jx == I) {
        if (MODE == 0) {
          const double x0[2] = {fp_from_u64(o.a0.x), fp_from_u64(o.a0.y)}, x1[2] = {fp_from_u64(o.a1.x), fp_from_u64(o.a1.y)};
          const double y0[2] = {fp_from_u64(o.b0.x), fp_from_u64(o.b0.y)}, y1[2] = {fp_from_u64(o.b1.x), fp_from_u64(o.b1.y)};
#pragma unroll
          for (int k = 0; k < 2; k++) {
            x[k] = fp_mulmod(x1[k], y1[k], q, qinv);
            d0[k] = fp_mulmod(x0[k], y0[k], q, qinv);
            d1[k] = fp_mulmod(x0[k], y1[k], q, qinv) + fp_mulmod(x1[k], y0[k], q, qinv);
          }
        } else {
#pragma unroll
          for (int k = 0; k < 2; k++) {
            x[k] = fp_from_u64(o.xs[k]);
            d0[k] = fp_from_u64(o.d0s[k]);
            d1[k] = fp_from_u64(o.d1s[k]);
          }
        }
      } else {
        const int w = Jx < I ? Jx : Jx - 1;
        const f64x2 v = *reinterpret_cast<const f64x2 *>(dyn + w * lds_words(10) + lds_pad(e));
        x[0] = v.x;
        x[1] = v.y;
      }
      s0[0] += fp_mulmod(x[0], kd(o.k0[Jx].x), q, qinv);
      s0[1] += fp_mulmod(x[1], kd(o.k0[Jx].y), q, qinv);
      s1[0] += fp_mulmod(x[0], kd(o.k1[Jx].x), q, qinv);
      s1[1] += fp_mulmod(x[1], kd(o.k1[Jx].y), q, qinv);
      if (Jx == 7) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
          s0[k] = fp_centre(s0[k], q, qinv);
          s1[k] = fp_centre(s1[k], q, qinv);
        }
      }
    }
    const f64x2 u0 = *reinterpret_cast<const f64x2 *>(tt0 + lds_pad(e)), u1 = *reinterpret_cast<const f64x2 *>(tt1 + lds_pad(e));
    u64x2 r;
    r.x = fp_to_canon(fp_mul_lazy(s0[0] - u0.x, inv, inv_q, q) + d0[0], q, qinv);
    r.y = fp_to_canon(fp_mul_lazy(s0[1] - u0.y, inv, inv_q, q) + d0[1], q, qinv);
    *reinterpret_cast<u64x2 *>(out + ((ct * 2 + 0) * nl + I) * N + base + e) = r;
    r.x = fp_to_canon(fp_mul_lazy(s1[0] - u1.x, inv, inv_q, q) + d1[0], q, qinv);
    r.y = fp_to_canon(fp_mul_lazy(s1[1] - u1.y, inv, inv_q, q) + d1[1], q, qinv);
    *reinterpret_cast<u64x2 *>(out + ((ct * 2 + 1) * nl + I) * N + base + e) = r;
  };
  if (keyf) compute_pair(2 * (int)threadIdx.x, ops, std::true_type{});  // workgroup-uniform
  else compute_pair(2 * (int)threadIdx.x, ops, std::false_type{});
}

template <int MODE, bool GAL>
static bool launch_split4_main(hipStream_t st, abc_hip_ctx *c, size_t cc, int nl, const double *part, const double *tpart, const u64 *opa,
                               const u64 *opb, size_t opa_stride, size_t opb_stride, int add_c1, const u64 *key, u64 *out, u32 gelt,
                               u32 imap = 0x76543210u, int ni = -1, int pack = 0) {
  if (nl < 1 || nl > 7) return false;  // nl + 1 transform buffers of 8.5 KiB + the table: two workgroups per CU up to nl = 6, one at 7
  if (ni < 0) ni = nl;
  if (ni == 0) return true;
  const dim3 grid((unsigned)(cc * ni * 16)), block(512);
  const size_t lds = (size_t)((nl + 1) * lds_words(10)) * 8 + 1024 * 16;
#define ABC_TM4(NLV)                                                                                                                 \
  hipLaunchKernelGGL((k_split4_main_fp<MODE, GAL, NLV>), grid, block, lds, st, c->dc, part, tpart, opa, opb, opa_stride, opb_stride, \
                     add_c1, key, out, gelt, imap, ni, pack, keyf)
  const double *keyf = key_twin_lookup(c, key);
  switch (nl) {
    case 1: ABC_TM4(1); break;
    case 2: ABC_TM4(2); break;
    case 3: ABC_TM4(3); break;
    case 4: ABC_TM4(4); break;
    case 5: ABC_TM4(5); break;
    case 6: ABC_TM4(6); break;
    default: ABC_TM4(7); break;
  }
#undef ABC_TM4
  return true;
}
// the same launch over a subset of the data primes (mixed chains, abc_kernels_isplit.hip): mode 0 multiply, mode 1 key switch
bool split4_main_subset(hipStream_t st, abc_hip_ctx *c, size_t cc, int nl, int mode, const double *part, const double *tpart, const u64 *opa,
                        const u64 *opb, size_t opa_stride, size_t opb_stride, int add_c1, const u64 *key, u64 *out, u32 gelt, u32 imap,
                        int ni) {
  if (mode == 0) return launch_split4_main<0, false>(st, c, cc, nl, part, tpart, opa, opb, opa_stride, opb_stride, add_c1, key, out, gelt, imap, ni);
  if (gelt) return launch_split4_main<1, true>(st, c, cc, nl, part, tpart, opa, opb, opa_stride, opb_stride, add_c1, key, out, gelt, imap, ni);
  return launch_split4_main<1, false>(st, c, cc, nl, part, tpart, opa, opb, opa_stride, opb_stride, add_c1, key, out, gelt, imap, ni);
}


// K2a..K2c on one chunk (the half-done decomposition limbs are in s.dec)
template <int MODE, bool GAL>
static void launch_split3(hipStream_t st, abc_hip_ctx *c, const FusedScratch &s, size_t cc, int nl, const u64 *opa, const u64 *opb,
                          size_t opa_stride, size_t opb_stride, int add_c1, const u64 *key, u64 *out, u32 gelt, int pack) {
  launch_split_special(st, c, cc, nl, (const double *)s.dec, key, (double *)s.tsp);
  hipLaunchKernelGGL(k_split3_pass_fp<14>, dim3((unsigned)(cc * 2 * 4)), dim3(256), 0, st, c->dc, (const double *)s.tsp, (double *)s.ksacc,
                     nl, pack);
  // (measured at nl = 5 / 6 / 7, every prime below 2^50: +9.5 / -5 / -14 % against k_split3_main_fp: above five limbs the prefetched
  // key words push the kernel past 128 VGPRs and to one workgroup per CU)
  if (!c->sw.no_split4 && nl <= 5 && launch_split4_main<MODE, GAL>(st, c, cc, nl, (const double *)s.dec, (const double *)s.ksacc, opa, opb,
                                                        opa_stride, opb_stride, add_c1, key, out, gelt, 0x76543210u, -1, pack))
    return;
  launch_split3_main<MODE, GAL>(st, c, cc, nl, (const double *)s.dec, (const double *)s.ksacc, opa, opb, opa_stride, opb_stride, add_c1,
                                key, out, gelt);
}

// packed half-done limbs (abc_ntt.hpp): only the sequence whose consumer is k_split4_main_fp reads them
static inline int pack_half_done(const abc_hip_ctx *c, int nl) {
  return (!c->sw.no_pack && !c->sw.no_split4 && nl >= 1 && nl <= 5) ? 1 : 0;
}
static inline bool all_fp(const abc_hip_ctx *c) {  // fp64 transforms: every key prime below 2^50
  bool fp = c->use_fp;
  for (int j = 0; j < c->K; j++) fp = fp && fp_ok(c->h_mods[j].bits);
  return fp;
}
static inline bool needs_guard(const abc_hip_ctx *c) {  // unguarded butterflies need (2 logN + 4) q < 2^64 for every key prime
  bool guard = false;
  for (int j = 0; j < c->K; j++) guard = guard || !unguarded_ok(c->h_mods[j].bits);
  return guard;
}

// K2a..K3 on one chunk: operand given by (coef, coef_stride) [+ (ntt, ntt_stride) for CKKS], addends by (addend, stride)
template <int LB>
static int keyswitch_stage(abc_hip_ctx *c, hipStream_t st, const FusedScratch &s, const u64 *coef, size_t coef_stride, const u64 *ntt,
                           size_t ntt_stride, const u64 *key, const u64 *addend, size_t addend_stride, bool add_c1, u64 *out, int nl,
                           size_t cc, int dec_ready = 0 /* 1: dec already holds the transformed decomposition limbs */) {
  const size_t N = (size_t)1 << LB;
  const dim3 block((1 << LB) / 16);
  const bool ckks = (c->scheme == 2);
  const bool guard = needs_guard(c);
  const unsigned g2a = (unsigned)(cc * (nl + 1) * nl);
  const unsigned g3 = (unsigned)(cc * 2 * nl);
  if (all_fp(c)) {
    if (!dec_ready)
      hipLaunchKernelGGL(k_fused_ks_decomp_ntt_fp<LB>, dim3(g2a), block, 0, st, c->dc, coef, coef_stride, s.dec, nl, ckks ? 1 : 0);
    hipLaunchKernelGGL(k_fused_ks_mac, dim3(stream_grid(cc * (nl + 1) * (N / 2), 256)), dim3(256), 0, st, c->dc, s.dec,
                       ckks ? ntt : (const u64 *)nullptr, ntt_stride, key, s.ksacc, s.tsp, nl, cc);
    hipLaunchKernelGGL(k_fused_ks_special_intt_fp<LB>, dim3((unsigned)(cc * 2)), block, 0, st, c->dc, s.tsp, s.tlast);
    if (ckks)
      hipLaunchKernelGGL((k_fused_ks_moddown_fp<LB, false>), dim3(g3), block, 0, st, c->dc, s.ksacc, s.tlast, addend, addend_stride,
                         add_c1 ? 1 : 0, out, nl, (int)(cc * 2), 0u);
    else
      hipLaunchKernelGGL(k_fused_ks_moddown_bfv_fp<LB>, dim3(g3), block, 0, st, c->dc, s.ksacc, s.tlast, addend, addend_stride,
                         add_c1 ? 1 : 0, out, nl);
    ABC_HIP_CHECK(hipGetLastError());
    return 0;
  }
  bool lazy = !guard;  // 4 products of a (< 64q) operand with a key residue must stay below 2^(k+63): k <= 55
  for (int j = 0; j < c->K; j++) lazy = lazy && c->h_mods[j].bits <= 55;
  if (guard)
    hipLaunchKernelGGL((k_fused_ks_decomp_ntt<LB, true, false>), dim3(g2a), block, 0, st, c->dc, coef, coef_stride, s.dec, nl,
                       ckks ? 1 : 0);
  else if (lazy)
    hipLaunchKernelGGL((k_fused_ks_decomp_ntt<LB, false, true>), dim3(g2a), block, 0, st, c->dc, coef, coef_stride, s.dec, nl,
                       ckks ? 1 : 0);
  else
    hipLaunchKernelGGL((k_fused_ks_decomp_ntt<LB, false, false>), dim3(g2a), block, 0, st, c->dc, coef, coef_stride, s.dec, nl,
                       ckks ? 1 : 0);
  hipLaunchKernelGGL(k_fused_ks_mac, dim3(stream_grid(cc * (nl + 1) * (N / 2), 256)), dim3(256), 0, st, c->dc, s.dec,
                     ckks ? ntt : (const u64 *)nullptr, ntt_stride, key, s.ksacc, s.tsp, nl, cc);
  hipLaunchKernelGGL(k_fused_ks_special_intt<LB>, dim3((unsigned)(cc * 2)), block, 0, st, c->dc, s.tsp, s.tlast);
  if (!ckks)
    hipLaunchKernelGGL(k_fused_ks_moddown_bfv<LB>, dim3(g3), block, 0, st, c->dc, s.ksacc, s.tlast, addend, addend_stride,
                       add_c1 ? 1 : 0, out, nl);
  else if (guard)
    hipLaunchKernelGGL((k_fused_ks_moddown<LB, true>), dim3(g3), block, 0, st, c->dc, s.ksacc, s.tlast, addend, addend_stride,
                       add_c1 ? 1 : 0, out, nl);
  else
    hipLaunchKernelGGL((k_fused_ks_moddown<LB, false>), dim3(g3), block, 0, st, c->dc, s.ksacc, s.tlast, addend, addend_stride,
                       add_c1 ? 1 : 0, out, nl);
  ABC_HIP_CHECK(hipGetLastError());
  return 0;
}

struct ChunkPlan {
  size_t chunk;
  int lanes;
};
// Chunks alternate between two internal streams so that the HBM-streaming kernels of one chunk overlap the ALU-bound
// transforms of the other; measured on MI355X: 256-pair chunks on two lanes beat one 512-pair chunk per lane.
static ChunkPlan plan_chunks(const abc_hip_ctx *c, int nl, size_t count) {
  ChunkPlan p{c->sw.chunk, c->sw.lanes};
  if (count <= 8) p.lanes = 1;
  if (!p.chunk) {
    const size_t per_ct_bytes = fused_scratch_limbs(nl) * c->n * 8;
    const size_t cap = ((size_t)4 << 30) / per_ct_bytes / (size_t)p.lanes;  // scratch capped at 4 GiB
    p.chunk = (count + p.lanes - 1) / p.lanes;
    if (p.chunk > 128) p.chunk = 128;  // measured this round: 128 > 256 > 64 > 512 (+2 / 0 / -0.5 / -1.5 %)
    if (p.chunk > cap) p.chunk = cap;
    if (p.chunk < 1) p.chunk = 1;
  }
  if (p.chunk > count) p.chunk = count;
  return p;
}

// one wavefront that sleeps for `ticks` of the 100 MHz wall clock (lane phase offset experiment)
__global__ void k_lane_delay(unsigned ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

int fork_lanes(abc_hip_ctx *c, int lanes) {
  if (lanes < 2) return 0;
  ABC_HIP_CHECK(hipEventRecord(c->lane_fork, c->stream));
  for (int l = 0; l < lanes; l++) ABC_HIP_CHECK(hipStreamWaitEvent(c->lane[l], c->lane_fork, 0));
  // ABC_HIP_LANE_OFFSET_US: lane l starts l x this late, so that the lanes sit in different kernels of the sequence
  if (const unsigned us = c->sw.lane_offset_us)
    for (int l = 1; l < lanes; l++) hipLaunchKernelGGL(k_lane_delay, dim3(1), dim3(64), 0, c->lane[l], us * 100u * l);
  return 0;
}
int join_lanes(abc_hip_ctx *c, int lanes) {
  if (lanes < 2) return 0;
  for (int l = 0; l < lanes; l++) {
    ABC_HIP_CHECK(hipEventRecord(c->lane_join[l], c->lane[l]));
    ABC_HIP_CHECK(hipStreamWaitEvent(c->stream, c->lane_join[l], 0));
  }
  return 0;
}

static int run_isplit(abc_hip_ctx *c, int mode, const u64 *opa, const u64 *opb, size_t opa_stride, size_t opb_stride, bool add_c1,
                      const u64 *key, u64 *out, int nl, size_t count, u32 gelt) {
  const size_t N = (size_t)c->n;
  const ChunkPlan p = plan_chunks(c, nl, count);
  const size_t per_ct = isplit_scratch_words(c, nl);
  if (ensure_workspace(c, (size_t)p.lanes * p.chunk * per_ct * 8)) return 1;
  if (c->logn == 14 && !c->sw.no_mixed) (void)key_twin(c, key);  // the fp64 limbs of a mixed chain go through k_split4_main_fp
  LaneScope scope(c, p.lanes);
  if (scope.fork()) return 1;
  int turn = 0;
  for (size_t off = 0; off < count; off += p.chunk, turn++) {
    const size_t cc = (count - off < p.chunk) ? count - off : p.chunk;
    const int l = (p.lanes > 1) ? turn % p.lanes : 0;
    hipStream_t st = (p.lanes > 1) ? c->lane[l] : c->stream;
    u64 *scratch = (u64 *)c->ws + (size_t)l * p.chunk * per_ct;
    if (isplit_chunk(c, st, scratch, cc, nl, mode, opa + off * opa_stride, opb ? opb + off * opb_stride : nullptr, opa_stride, opb_stride,
                     add_c1 ? 1 : 0, key, out + off * 2 * (size_t)nl * N, gelt))
      return 1;
  }
  return scope.join();
}

// N = 2^15 (abc_kernels_gsplit.hip)
static int run_gsplit15(abc_hip_ctx *c, int mode, const u64 *opa, const u64 *opb, size_t opa_stride, size_t opb_stride, bool add_c1,
                        const u64 *key, u64 *out, int nl, size_t count, u32 gelt) {
  const size_t N = (size_t)c->n;
  ChunkPlan p = plan_chunks(c, nl, count);
  const size_t per_ct = gsplit_scratch_words(c, nl);
  if (ensure_workspace(c, (size_t)p.lanes * p.chunk * per_ct * 8)) return 1;
  LaneScope scope(c, p.lanes);
  if (scope.fork()) return 1;
  int turn = 0;
  for (size_t off = 0; off < count; off += p.chunk, turn++) {
    const size_t cc = (count - off < p.chunk) ? count - off : p.chunk;
    const int l = (p.lanes > 1) ? turn % p.lanes : 0;
    hipStream_t st = (p.lanes > 1) ? c->lane[l] : c->stream;
    u64 *scratch = (u64 *)c->ws + (size_t)l * p.chunk * per_ct;
    if (gsplit_chunk15(c, st, scratch, cc, nl, mode, opa + off * opa_stride, opb ? opb + off * opb_stride : nullptr, opa_stride,
                       opb_stride, add_c1 ? 1 : 0, key, out + off * 2 * (size_t)nl * N, gelt))
      return 1;
  }
  return scope.join();
}

// ---- CKKS multiply + relinearise ----
// integer split sequence (abc_kernels_isplit.hip): chains with a prime above 2^50
static int run_isplit(abc_hip_ctx *c, int mode, const u64 *opa, const u64 *opb, size_t opa_stride, size_t opb_stride, bool add_c1,
                      const u64 *key, u64 *out, int nl, size_t count, u32 gelt);

template <int LB>
static int run_mul_relin(abc_hip_ctx *c, const u64 *a, const u64 *b, u64 *out, int nl, size_t count) {
  const size_t N = (size_t)1 << LB;
  if (LB == 14 && !all_fp(c) && isplit_applies(c, nl))
    return run_isplit(c, 0, a, b, 2 * (size_t)nl * N, 2 * (size_t)nl * N, false, c->d_relin, out, nl, count, 0u);
  const ChunkPlan p = plan_chunks(c, nl, count);
  // the split sequence (N = 2^14, every key prime below 2^50); its scratch limbs are c->dc.ps words apart (N plus an optional
  // pad, see abc_hip_ctx_create).  Otherwise: the LDS-resident kernels (smaller rings, ABC_HIP_NO_SPLIT, wider primes)
  const bool split = LB == 14 && all_fp(c) && !c->sw.no_split && nl <= 12;
  const size_t SN = split ? (size_t)c->dc.ps : N;
  const size_t per_ct = fused_scratch_limbs(nl) * SN;
  if (ensure_workspace(c, (size_t)p.lanes * p.chunk * per_ct * 8)) return 1;
  if (split) (void)key_twin(c, c->d_relin);  // before the lanes fork: they order themselves behind c->stream
  LaneScope scope(c, p.lanes);
  if (scope.fork()) return 1;
  const size_t ctw = 2 * (size_t)nl * N;
  int turn = 0;
  for (size_t off = 0; off < count; off += p.chunk, turn++) {
    const size_t cc = (count - off < p.chunk) ? count - off : p.chunk;
    const int l = (p.lanes > 1) ? turn % p.lanes : 0;
    hipStream_t st = (p.lanes > 1) ? c->lane[l] : c->stream;
    const FusedScratch s = carve((u64 *)c->ws + (size_t)l * p.chunk * per_ct, p.chunk, nl, SN);
    if constexpr (LB == 14) {
      if (split) {
        // few ciphertexts in flight: the 139 KiB workgroups of the tensor kernel would leave most CUs idle for its whole
        // duration; the block-wise inverse tails + register cross pass of abc_kernels_gsplit.hip spread over the chip instead
        const bool lean = !c->sw.no_lean_front && cc * nl <= c->sw.lean_limit;  // measured at nl = 4: +5 % at 16 pairs, even at 32, -5 % at 48
        const int pack = pack_half_done(c, nl);
        if (lean)
          gsplit_front14(st, c, cc, nl, 0, a + off * ctw, b + off * ctw, 0, (double *)s.coef, (double *)s.dec, 0u, pack);
        else
          hipLaunchKernelGGL(k_split2_tensor_pass0_fp<LB>, dim3((unsigned)(cc * nl)), dim3((1 << LB) / 16), 0, st, c->dc, a + off * ctw,
                             b + off * ctw, (double *)s.dec, nl, pack);
        launch_split3<0, false>(st, c, s, cc, nl, a + off * ctw, b + off * ctw, 0, 0, 0, c->d_relin, out + off * ctw, 0u, pack);
        ABC_HIP_CHECK(hipGetLastError());
        continue;
      }
    }
    if (all_fp(c))  // tensor product, inverse transform and the forward transforms of the decomposition in one LDS-resident kernel
      hipLaunchKernelGGL(k_fused_tensor_decomp_fp<LB>, dim3((unsigned)(cc * nl)), dim3((1 << LB) / 16), 0, st, c->dc, a + off * ctw,
                         b + off * ctw, s.c01, s.ntt, s.dec, nl);
    else
      hipLaunchKernelGGL(k_fused_tensor_intt<LB>, dim3((unsigned)(cc * nl)), dim3((1 << LB) / 16), 0, st, c->dc, a + off * ctw,
                         b + off * ctw, s.c01, s.coef, s.ntt, nl);
    if (keyswitch_stage<LB>(c, st, s, s.coef, (size_t)nl * N, s.ntt, (size_t)nl * N, c->d_relin, s.c01, ctw, true, out + off * ctw,
                            nl, cc, all_fp(c) ? 1 : 0))
      return 1;
  }
  return scope.join();
}

// -1: not applicable (ring too large for an LDS-resident limb) -> caller takes the generic path
int ckks_mul_relin_fused(abc_hip_ctx *c, const u64 *a, const u64 *b, u64 *out, int nl, size_t count) {
  if (c->logn == 15 && gsplit_applies(c, nl)) {
    if (!count) return 0;
    const size_t ctw = 2 * (size_t)nl * c->n;
    return run_gsplit15(c, 0, a, b, ctw, ctw, false, c->d_relin, out, nl, count, 0u);
  }
  if (c->logn == 15 && !all_fp(c) && isplit_applies(c, nl)) {  // a prime above 2^50: the integer split sequence with 32 blocks
    if (!count) return 0;
    const size_t ctw = 2 * (size_t)nl * c->n;
    return run_isplit(c, 0, a, b, ctw, ctw, false, c->d_relin, out, nl, count, 0u);
  }
  if (c->logn > 14) return -1;
  if (c->sw.no_fused) return -1;
  if (!count) return 0;
  switch (c->logn) {
    case 10: return run_mul_relin<10>(c, a, b, out, nl, count);
    case 11: return run_mul_relin<11>(c, a, b, out, nl, count);
    case 12: return run_mul_relin<12>(c, a, b, out, nl, count);
    case 13: return run_mul_relin<13>(c, a, b, out, nl, count);
    case 14: return run_mul_relin<14>(c, a, b, out, nl, count);
    default: return -1;
  }
}

// ---- general key switch: out[ct] = KeySwitch(target[ct]) (+ addend) ----
// target: [nl][N] per ciphertext at target + ct*target_stride, in the ciphertext's own form (BFV coefficient, CKKS NTT)
template <int LB>
static int run_keyswitch(abc_hip_ctx *c, const u64 *target, size_t target_stride, const u64 *key, u64 *out, int nl, size_t count,
                         const u64 *addend, size_t addend_stride, bool add_c1, u32 gelt = 0) {
  const size_t N = (size_t)1 << LB;
  const bool ckks = (c->scheme == 2);
  if (LB == 14 && ckks && !all_fp(c) && isplit_applies(c, nl))
    return run_isplit(c, 1, target, addend, target_stride, addend_stride, add_c1, key, out, nl, count, gelt);
  const ChunkPlan p = plan_chunks(c, nl, count);
  // split sequences (N = 2^14, every key prime below 2^50): CKKS as in run_mul_relin; BFV (coefficient-form operand) the
  // register pass + abc_kernels_gsplit.hip's k_gsplit_special<14, NL, true> / k_bsplit_tcoef / k_bsplit_finish_big
  const bool splitc = LB == 14 && ckks && all_fp(c) && !c->sw.no_split && nl <= 12;
  const bool splitb = LB == 14 && !ckks && !c->sw.no_split && bsplit_applies(c, nl);
  // BFV rotation (coefficient form): the kernels gather with elt^-1 mod 2N
  const u32 ginv = (!ckks && gelt) ? (u32)host::invmod(gelt, 2 * (uint64_t)N) : 0u;
  if (ginv && !splitb) { set_error("run_keyswitch: a BFV permutation is only folded into the split sequence"); return 1; }
  const size_t SN = (splitc || splitb) ? (size_t)c->dc.ps : N;
  const size_t per_ct = fused_scratch_limbs(nl) * SN;
  if (ensure_workspace(c, (size_t)p.lanes * p.chunk * per_ct * 8)) return 1;
  if (splitc || splitb) (void)key_twin(c, key);  // before the lanes fork (BFV: the inner-product kernel reads it)
  LaneScope scope(c, p.lanes);
  if (scope.fork()) return 1;
  int turn = 0;
  for (size_t off = 0; off < count; off += p.chunk, turn++) {
    const size_t cc = (count - off < p.chunk) ? count - off : p.chunk;
    const int l = (p.lanes > 1) ? turn % p.lanes : 0;
    hipStream_t st = (p.lanes > 1) ? c->lane[l] : c->stream;
    const FusedScratch s = carve((u64 *)c->ws + (size_t)l * p.chunk * per_ct, p.chunk, nl, SN);
    const u64 *tg = target + off * target_stride;
    const u64 *ad = addend ? addend + off * addend_stride : nullptr;
    u64 *o = out + off * 2 * nl * N;
    if constexpr (LB == 14) {
      if (splitc) {
        const int pack = pack_half_done(c, nl);
        if (!c->sw.no_lean_front && cc * nl <= c->sw.lean_limit)
          gsplit_front14(st, c, cc, nl, 1, tg, nullptr, target_stride, (double *)s.coef, (double *)s.dec, gelt, pack);
        else
          hipLaunchKernelGGL((gelt ? k_fused_operand_pass0_fp<LB, true, true> : k_fused_operand_pass0_fp<LB, true, false>),
                             dim3((unsigned)(cc * nl)), dim3((1 << LB) / 16), 0, st, c->dc, tg, target_stride, (double *)s.dec, nl, 0,
                             gelt, 1 | (pack ? 2 : 0));
        if (gelt)
          launch_split3<1, true>(st, c, s, cc, nl, tg, ad, target_stride, addend_stride, add_c1 ? 1 : 0, key, o, gelt, pack);
        else
          launch_split3<1, false>(st, c, s, cc, nl, tg, ad, target_stride, addend_stride, add_c1 ? 1 : 0, key, o, 0u, pack);
        ABC_HIP_CHECK(hipGetLastError());
        continue;
      }
      if (splitb) {
        // few ciphertexts in flight: one workgroup per (ct, J, target I) instead of per (ct, J)
        const bool per_target = cc * nl < c->sw.pass0_target_limit;
        hipLaunchKernelGGL((k_fused_operand_pass0_fp<LB, false, false>), dim3((unsigned)(cc * nl * (per_target ? nl + 1 : 1))),
                           dim3((1 << LB) / 16), 0, st, c->dc, tg, target_stride, (double *)s.dec, nl, per_target ? 1 : 0, ginv, 1);
        // inner product + inverse tails for every key prime, then the register-only finish (a rotation's addend g(c0): gathered there)
        if (bsplit_back14(c, st, cc, nl, (const double *)s.dec, (double *)s.ksacc, key, ad, addend_stride, add_c1 ? 1 : 0, o, ginv)) return 1;
        continue;
      }
    }
    // LDS-resident kernels (smaller rings, wider primes, ABC_HIP_NO_SPLIT)
    const u64 *coef = tg;
    size_t coef_stride = target_stride;
    if (ckks) {  // operand arrives in NTT form: coefficient form via one in-LDS inverse transform per limb
      if (all_fp(c))
        hipLaunchKernelGGL(k_fused_operand_intt_fp<LB>, dim3((unsigned)(cc * nl)), dim3((1 << LB) / 16), 0, st, c->dc, tg,
                           target_stride, s.coef, nl);
      else
        hipLaunchKernelGGL(k_fused_operand_intt<LB>, dim3((unsigned)(cc * nl)), dim3((1 << LB) / 16), 0, st, c->dc, tg, target_stride,
                           s.coef, nl);
      coef = s.coef;
      coef_stride = (size_t)nl * N;
    }
    if (keyswitch_stage<LB>(c, st, s, coef, coef_stride, tg, target_stride, key, ad, addend_stride, add_c1, o, nl, cc, 0)) return 1;
  }
  return scope.join();
}

// CKKS rotation with the Galois permutation folded into the key switch (N = 2^14, fp64 split path): in [count][2][nl][N]
// NTT form; out = (g(c0) + ks0, ks1), ks = KeySwitch(g(c1)).  -1: not applicable, caller permutes first.
int rotate_fused(abc_hip_ctx *c, const u64 *in, u32 elt, const u64 *key, u64 *out, int nl, size_t count) {
  if (c->logn == 15 && in != out && gsplit_applies(c, nl) && !c->sw.no_galois_fusion) {
    if (!count) return 0;
    const size_t pw15 = (size_t)nl * c->n;
    return run_gsplit15(c, 1, in + pw15, in, 2 * pw15, 2 * pw15, false, key, out, nl, count, elt);
  }
  if (c->logn == 15 && c->scheme == 2 && in != out && !all_fp(c) && isplit_applies(c, nl) && !c->sw.no_galois_fusion) {
    if (!count) return 0;
    const size_t pw15 = (size_t)nl * c->n;
    return run_isplit(c, 1, in + pw15, in, 2 * pw15, 2 * pw15, false, key, out, nl, count, elt);
  }
  if (c->scheme == 1 && in != out && !c->sw.no_galois_fusion && !c->sw.no_split && !c->sw.no_fused) {
    // BFV, coefficient form: the signed permutation folded into the first step's load and the last step's addend (fp64 chains)
    const size_t pwb = (size_t)nl * c->n;
    if (c->logn == 14 && bsplit_applies(c, nl)) {
      if (!count) return 0;
      return run_keyswitch<14>(c, in + pwb, 2 * pwb, key, out, nl, count, in, 2 * pwb, false, elt);
    }
    if ((c->logn == 13 || ((c->logn == 15 || c->logn == 16) && !c->sw.no_finish_lds)) && bsplit_big_applies(c, nl)) {
      if (!count) return 0;
      return bsplit_big(c, in + pwb, 2 * pwb, key, out, nl, count, in, 2 * pwb, false, (u32)host::invmod(elt, 2 * (uint64_t)c->n));
    }
  }
  if (c->scheme == 1 && in != out && !c->sw.no_galois_fusion && iks_bfv_applies(c, nl) && !bsplit_big_applies(c, nl)) {
    if (!count) return 0;  // BFV on a big ring with a prime above 2^50: k_iks_pass0 / k_iks_finish gather
    const size_t pwb = (size_t)nl * c->n;
    return keyswitch_generic(c, in + pwb, 2 * pwb, key, out, nl, count, in, 2 * pwb, false, (u32)host::invmod(elt, 2 * (uint64_t)c->n));
  }
  if (c->logn != 14 || c->scheme != 2 || in == out) return -1;
  if (!all_fp(c) && !isplit_applies(c, nl)) return -1;
  if (c->sw.no_split || c->sw.no_fused || c->sw.no_galois_fusion) return -1;
  if (!count) return 0;
  const size_t N = (size_t)c->n, pw = (size_t)nl * N;
  return run_keyswitch<14>(c, in + pw, 2 * pw, key, out, nl, count, in, 2 * pw, false, elt);
}

int keyswitch_fused(abc_hip_ctx *c, const u64 *target, size_t target_stride, const u64 *key, u64 *out, int nl, size_t count,
                    const u64 *addend, size_t addend_stride, bool add_c1) {
  if (c->logn == 15 && gsplit_applies(c, nl)) {
    if (!count) return 0;
    return run_gsplit15(c, 1, target, addend, target_stride, addend_stride, add_c1, key, out, nl, count, 0u);
  }
  if (c->logn == 15 && c->scheme == 2 && !all_fp(c) && isplit_applies(c, nl)) {
    if (!count) return 0;
    return run_isplit(c, 1, target, addend, target_stride, addend_stride, add_c1, key, out, nl, count, 0u);
  }
  if (bsplit_big_applies(c, nl)) {  // BFV, N = 2^15 / 2^16, fp64-capable chain
    if (!count) return 0;
    return bsplit_big(c, target, target_stride, key, out, nl, count, addend, addend_stride, add_c1);
  }
  if (c->logn > 14) return -1;
  if (c->sw.no_fused) return -1;
  if (!count) return 0;
  switch (c->logn) {
    case 10: return run_keyswitch<10>(c, target, target_stride, key, out, nl, count, addend, addend_stride, add_c1);
    case 11: return run_keyswitch<11>(c, target, target_stride, key, out, nl, count, addend, addend_stride, add_c1);
    case 12: return run_keyswitch<12>(c, target, target_stride, key, out, nl, count, addend, addend_stride, add_c1);
    case 13: return run_keyswitch<13>(c, target, target_stride, key, out, nl, count, addend, addend_stride, add_c1);
    case 14: return run_keyswitch<14>(c, target, target_stride, key, out, nl, count, addend, addend_stride, add_c1);
    default: return -1;
  }
}

}  // namespace abc
