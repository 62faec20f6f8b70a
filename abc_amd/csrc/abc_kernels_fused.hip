// abc_kernels_fused.hip -- the hot path: CKKS ciphertext x ciphertext multiply + relinearise in five
// launches (N <= 2^14, one workgroup per RNS limb, the limb resident in LDS).
//
// Replaces SealCiphertext::multiply / multiplyInplace = Evaluator::multiply + relinearize_inplace
// (src/runtime/SealCiphertext.cpp:102-107,121-124) for the CKKS scheme north_star names.
//   K1 tensor_intt : c0 = a0b0, c1 = a0b1 + a1b0 written to scratch (so `out` may alias an operand); c2 = a1b1 kept in NTT form and,
//                    through an in-LDS inverse transform, in coefficient form (key-switch operand).
//   K2a decomp_ntt : workgroup (ct, I, J) reduces c2_J modulo key prime I and transforms it in LDS.
//   K2b mac        : streaming inner product with relin_key[J][.][I] (128-bit lazy accumulation).
//   K2c special    : the special-prime limb goes back to coefficients, plus the q_sp/2 rounding offset.
//                    (A single-kernel accumulate-in-registers form of K2a+K2b was measured first: 64 VGPRs
//                    of accumulators beside a 16-coefficient-per-lane transform exceed the 128-VGPR budget
//                    of a 1024-thread workgroup and spill; see DESIGN.md "Key switch: what was tried".)
//   K3 ks_moddown  : workgroup (ct, comp, j) reduces the special-prime polynomial modulo q_j, transforms
//                    it, subtracts, scales by q_sp^-1 and adds c0 / c1.
// Algorithmic HBM bytes per multiply: 8N(6L + 2L(L+1)); scratch per ciphertext:
// (6L+2) limbs.
#include <cstdlib>

#include "abc_context.hpp"

namespace abc {

// slot r = 4g+k of the transforms' final register layout  <->  element 4*(tid + T*g) + k
template <int LB>
__device__ __forceinline__ int slot_elem(int r) {
  return ((threadIdx.x + ((1 << LB) / 16) * (r >> 2)) << 2) + (r & 3);
}

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

// K2a: one workgroup per (ct, key prime I, decomposition limb J != I): residues of c2_J modulo key prime I,
// transformed in LDS -> dec[ct][I][J]
template <int LB, bool GUARD>
__global__ __launch_bounds__((1 << LB) / 16) void k_fused_ks_decomp_ntt(DevCtx c, const u64 *__restrict__ c2coef,
                                                                        u64 *__restrict__ dec, int nl) {
  __shared__ u64 lds[lds_words(LB)];
  const int J = blockIdx.x % nl;
  const int I = (blockIdx.x / nl) % (nl + 1);
  const size_t ct = blockIdx.x / ((size_t)nl * (nl + 1));
  if (J == I) return;  // q_J is the key prime itself: the NTT-form limb is used directly (CKKS branch)
  const size_t N = (size_t)1 << LB;
  const int ki = (I == nl) ? c.K - 1 : I;
  const Mod m = c.mods[ki];
  const NttTable t = ntt_table(c, ki);
  const u64 *__restrict__ src = c2coef + (ct * nl + J) * N;
  u64 *__restrict__ dst = dec + ((ct * (nl + 1) + I) * nl + J) * N;
  const bool need_reduce = c.mods[J].q > m.q;  // values are < q_J already
  ntt_fwd_block<LB, GUARD>(
      lds, [&](int, int i) { const u64 v = src[i]; return need_reduce ? reduce64(v, m) : v; },
      [&](int, int i, u64 v) { dst[i] = canon_fwd<GUARD>(v, m); }, t, m, 0, 0);
}

// K2b: streaming inner product with the key: acc_comp[k] = sum_J x_J[k] * key[J][comp][I][k]
__global__ __launch_bounds__(256) void k_fused_ks_mac(DevCtx c, const u64 *__restrict__ dec, const u64 *__restrict__ c2ntt,
                                                      const u64 *__restrict__ key, u64 *__restrict__ ksacc,
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
      const u64 *xs = (J == I) ? c2ntt + (ct * nl + J) * N + k : dec + ((ct * (nl + 1) + I) * nl + J) * N + k;
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

template <int LB, bool GUARD>
__global__ __launch_bounds__((1 << LB) / 16) void k_fused_ks_moddown(DevCtx c, const u64 *__restrict__ ksacc,
                                                                     const u64 *__restrict__ tlast, const u64 *__restrict__ c01,
                                                                     u64 *__restrict__ out, int nl) {
  __shared__ u64 lds[lds_words(LB)];
  const int j = blockIdx.x % nl;
  const size_t cc = blockIdx.x / nl;  // ct*2 + comp
  const size_t N = (size_t)1 << LB;
  const Mod m = c.mods[j];
  const NttTable t = ntt_table(c, j);
  const u64 half = c.mods[c.K - 1].q >> 1;
  const u64 hm = reduce64(half, m);
  const u64 fix = hm ? m.q - hm : 0;
  const u64 inv = c.cst->inv_special[j], inv_s = c.cst->inv_special_s[j];
  const u64 *__restrict__ src = tlast + cc * N;
  const u64 *__restrict__ ks = ksacc + (cc * nl + j) * N;
  const u64 *__restrict__ cin = c01 + (cc * nl + j) * N;
  u64 *__restrict__ o = out + (cc * nl + j) * N;
  ntt_fwd_block<LB, GUARD>(
      lds, [&](int, int i) { return add_mod(reduce64(src[i], m), fix, m.q); },
      [&](int, int i, u64 v) {
        const u64 x = canon_fwd<GUARD>(v, m);
        o[i] = add_mod(mul_shoup(sub_mod(ks[i], x, m.q), inv, inv_s, m.q), cin[i], m.q);
      },
      t, m, 0, 0);
}

static inline unsigned stream_grid(size_t items, int block) {
  size_t g = (items + block - 1) / block;
  const size_t cap = 256 * 8 * 4;
  return (unsigned)(g < cap ? (g ? g : 1) : cap);
}

// scratch limbs per ciphertext: c2coef L, c2ntt L, dec L(L+1), ksacc 2L, tsp 2, tlast 2, c01 2L
static inline size_t fused_scratch_limbs(int nl) { return (size_t)nl * (nl + 1) + 6 * (size_t)nl + 4; }

template <int LB>
static int run_fused(abc_hip_ctx *c, const u64 *a, const u64 *b, u64 *out, int nl, size_t count, size_t chunk, int lanes) {
  const size_t N = (size_t)1 << LB;
  const size_t per_ct = fused_scratch_limbs(nl) * N;
  if (ensure_workspace(c, (size_t)lanes * chunk * per_ct * 8)) return 1;
  const dim3 block((1 << LB) / 16);
  bool guard = false;  // unguarded butterflies need (2 logN + 4) q < 2^64 for every key-level prime
  for (int j = 0; j < c->K; j++) guard = guard || !unguarded_ok(c->h_mods[j].bits);
  if (lanes > 1) {  // fork: the lanes start after everything already queued on the caller's stream
    ABC_HIP_CHECK(hipEventRecord(c->lane_fork, c->stream));
    for (int l = 0; l < lanes; l++) ABC_HIP_CHECK(hipStreamWaitEvent(c->lane[l], c->lane_fork, 0));
  }
  int turn = 0;
  for (size_t off = 0; off < count; off += chunk, turn++) {
    const size_t cc = (count - off < chunk) ? count - off : chunk;
    const int l = (lanes > 1) ? turn % lanes : 0;
    hipStream_t st = (lanes > 1) ? c->lane[l] : c->stream;
    u64 *base = (u64 *)c->ws + (size_t)l * chunk * per_ct;
    u64 *c2coef = base, *c2ntt = c2coef + chunk * nl * N;
    u64 *dec = c2ntt + chunk * nl * N;
    u64 *ksacc = dec + chunk * nl * (nl + 1) * N, *tsp = ksacc + chunk * 2 * nl * N;
    u64 *tlast = tsp + chunk * 2 * N, *c01 = tlast + chunk * 2 * N;
    const size_t ctw = 2 * (size_t)nl * N;
    hipLaunchKernelGGL(k_fused_tensor_intt<LB>, dim3((unsigned)(cc * nl)), block, 0, st, c->dc, a + off * ctw, b + off * ctw, c01,
                       c2coef, c2ntt, nl);
    if (guard)
      hipLaunchKernelGGL((k_fused_ks_decomp_ntt<LB, true>), dim3((unsigned)(cc * (nl + 1) * nl)), block, 0, st, c->dc, c2coef, dec, nl);
    else
      hipLaunchKernelGGL((k_fused_ks_decomp_ntt<LB, false>), dim3((unsigned)(cc * (nl + 1) * nl)), block, 0, st, c->dc, c2coef, dec, nl);
    hipLaunchKernelGGL(k_fused_ks_mac, dim3(stream_grid(cc * (nl + 1) * (N / 2), 256)), dim3(256), 0, st, c->dc, dec, c2ntt,
                       c->d_relin, ksacc, tsp, nl, cc);
    hipLaunchKernelGGL(k_fused_ks_special_intt<LB>, dim3((unsigned)(cc * 2)), block, 0, st, c->dc, tsp, tlast);
    if (guard)
      hipLaunchKernelGGL((k_fused_ks_moddown<LB, true>), dim3((unsigned)(cc * 2 * nl)), block, 0, st, c->dc, ksacc, tlast, c01,
                         out + off * ctw, nl);
    else
      hipLaunchKernelGGL((k_fused_ks_moddown<LB, false>), dim3((unsigned)(cc * 2 * nl)), block, 0, st, c->dc, ksacc, tlast, c01,
                         out + off * ctw, nl);
    ABC_HIP_CHECK(hipGetLastError());
  }
  if (lanes > 1) {  // join
    for (int l = 0; l < lanes; l++) {
      ABC_HIP_CHECK(hipEventRecord(c->lane_join[l], c->lane[l]));
      ABC_HIP_CHECK(hipStreamWaitEvent(c->stream, c->lane_join[l], 0));
    }
  }
  return 0;
}

// -1: not applicable (ring too large for an LDS-resident limb) -> caller takes the generic path
int ckks_mul_relin_fused(abc_hip_ctx *c, const u64 *a, const u64 *b, u64 *out, int nl, size_t count) {
  if (c->logn > 14) return -1;
  if (const char *e = std::getenv("ABC_HIP_NO_FUSED"))
    if (e[0] == '1') return -1;
  if (!count) return 0;
  // chunk: scratch is L(L+1)+6L+4 limbs per ciphertext.  Large grids matter more than cache residency here
  // (one 1024-thread workgroup per CU, so a launch needs >> 256 workgroups): take the whole batch,
  // capped at 4 GiB of scratch.
  size_t chunk = 0;
  int lanes = 2;
  if (const char *e = std::getenv("ABC_HIP_CHUNK")) chunk = (size_t)std::atol(e);
  if (const char *e = std::getenv("ABC_HIP_LANES")) lanes = std::atoi(e) >= 2 ? 2 : 1;
  if (!chunk) {
    // every chunk should still be several full waves of 256 workgroups; scratch capped at 4 GiB
    const size_t per_ct_bytes = fused_scratch_limbs(nl) * c->n * 8;
    const size_t cap = ((size_t)4 << 30) / per_ct_bytes / (size_t)lanes;
    chunk = (count + lanes - 1) / lanes;
    if (chunk > 256) chunk = 256;  // measured on MI355X: 256-pair chunks on two lanes beat one 512-pair chunk per lane
    if (chunk > cap) chunk = cap;
    if (chunk < 1) chunk = 1;
  }
  if (chunk > count) chunk = count;
  if (count <= 8) lanes = 1;
  switch (c->logn) {
    case 10: return run_fused<10>(c, a, b, out, nl, count, chunk, lanes);
    case 11: return run_fused<11>(c, a, b, out, nl, count, chunk, lanes);
    case 12: return run_fused<12>(c, a, b, out, nl, count, chunk, lanes);
    case 13: return run_fused<13>(c, a, b, out, nl, count, chunk, lanes);
    case 14: return run_fused<14>(c, a, b, out, nl, count, chunk, lanes);
    default: return -1;
  }
}

}  // namespace abc
