// abc_kernels_fused.hip -- the hot path: CKKS ciphertext x ciphertext multiply + relinearise in three
// fused launches (N <= 2^14, one workgroup per RNS limb, the limb resident in LDS).
//
// Replaces SealCiphertext::multiply / multiplyInplace = Evaluator::multiply + relinearize_inplace
// (src/runtime/SealCiphertext.cpp:102-107,121-124) for the CKKS scheme north_star names.
//   K1 tensor_intt : c0 = a0b0, c1 = a0b1 + a1b0 written to `out`; c2 = a1b1 kept in NTT form and,
//                    through an in-LDS inverse transform, in coefficient form (key-switch operand).
//   K2 ks_accum    : workgroup (ct, I) forms for every decomposition limb J the residues of c2_J modulo
//                    key prime I, transforms them in LDS and multiply-accumulates with both components of
//                    relin_key[J][.][I] in registers -- the L*(L+1) temporaries of SEAL's
//                    switch_key_inplace never exist in HBM.  The special-prime workgroup finishes with
//                    two inverse transforms and the +q_sp/2 rounding offset.
//   K3 ks_moddown  : workgroup (ct, comp, j) reduces the special-prime polynomial modulo q_j, transforms
//                    it, subtracts, scales by q_sp^-1 and adds c0 / c1.
// Algorithmic HBM bytes per multiply: 8N(6L + 2L(L+1)); scratch per ciphertext:
// (4L+2) limbs, sized so a chunk stays inside the 256 MiB Infinity Cache.
#include <cstdlib>

#include "abc_context.hpp"

namespace abc {

// slot r = 4g+k of the transforms' final register layout  <->  element 4*(tid + T*g) + k
template <int LB>
__device__ __forceinline__ int slot_elem(int r) {
  return ((threadIdx.x + ((1 << LB) / 16) * (r >> 2)) << 2) + (r & 3);
}

template <int LB>
__global__ __launch_bounds__((1 << LB) / 16) void k_fused_tensor_intt(DevCtx c, const u64 *a, const u64 *b, u64 *out, u64 *c2coef,
                                                                      u64 *c2ntt, int nl) {
  __shared__ u64 lds[lds_words(LB)];
  const int j = blockIdx.x % nl;
  const size_t ct = blockIdx.x / nl;
  const size_t N = (size_t)1 << LB, pw = (size_t)nl * N;
  const Mod m = c.mods[j];
  const NttTable t = ntt_table(c, j);
  const u64 *a0 = a + ct * 2 * pw + j * N, *a1 = a0 + pw;
  const u64 *b0 = b + ct * 2 * pw + j * N, *b1 = b0 + pw;
  u64 *o0 = out + ct * 2 * pw + j * N, *o1 = o0 + pw;
  u64 *dcoef = c2coef + (ct * nl + j) * N, *dntt = c2ntt + (ct * nl + j) * N;
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

template <int LB>
__global__ __launch_bounds__((1 << LB) / 16) void k_fused_ks_accum(DevCtx c, const u64 *c2coef, const u64 *c2ntt, const u64 *key,
                                                                   u64 *ksacc, u64 *tlast, int nl) {
  __shared__ u64 lds[lds_words(LB)];
  const int I = blockIdx.x % (nl + 1);
  const size_t ct = blockIdx.x / (nl + 1);
  const size_t N = (size_t)1 << LB;
  const int ki = (I == nl) ? c.K - 1 : I;
  const Mod m = c.mods[ki];
  const NttTable t = ntt_table(c, ki);
  u64 acc0[16], acc1[16];
#pragma unroll
  for (int r = 0; r < 16; r++) acc0[r] = acc1[r] = 0;
  for (int J = 0; J < nl; J++) {
    const u64 *k0 = key + (((size_t)J * 2 + 0) * c.K + ki) * N;
    const u64 *k1 = key + (((size_t)J * 2 + 1) * c.K + ki) * N;
    if (J == I) {
      // q_J == key prime: the NTT-form limb is the operand itself (switch_key_inplace, CKKS branch)
      const u64 *src = c2ntt + (ct * nl + J) * N;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int i = slot_elem<LB>(r);
        const u64 x = src[i];
        acc0[r] = add_mod(acc0[r], mul_mod(x, k0[i], m), m.q);
        acc1[r] = add_mod(acc1[r], mul_mod(x, k1[i], m), m.q);
      }
    } else {
      const u64 *src = c2coef + (ct * nl + J) * N;
      ntt_fwd_block<LB>(
          lds, [&](int, int i) { return reduce64(src[i], m); },
          [&](int r, int i, u64 v) {
            const u64 x = canon4(v, m);
            acc0[r] = add_mod(acc0[r], mul_mod(x, k0[i], m), m.q);
            acc1[r] = add_mod(acc1[r], mul_mod(x, k1[i], m), m.q);
          },
          t, m, 0, 0);
      __syncthreads();  // LDS is reused by the next decomposition limb
    }
  }
  if (I < nl) {
    u64 *d0 = ksacc + ((ct * 2 + 0) * nl + I) * N, *d1 = ksacc + ((ct * 2 + 1) * nl + I) * N;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int i = slot_elem<LB>(r);
      d0[i] = acc0[r];
      d1[i] = acc1[r];
    }
  } else {
    // special prime: back to coefficients and add q_sp/2 (rounding of the division by q_sp)
    const u64 half = m.q >> 1;
    u64 *d0 = tlast + (ct * 2 + 0) * N, *d1 = tlast + (ct * 2 + 1) * N;
    ntt_inv_block<LB>(
        lds, [&](int r, int) { return acc0[r]; }, [&](int, int i, u64 v) { d0[i] = add_mod(scale_inv_n(v, m), half, m.q); }, t,
        m, 0, 0);
    __syncthreads();
    ntt_inv_block<LB>(
        lds, [&](int r, int) { return acc1[r]; }, [&](int, int i, u64 v) { d1[i] = add_mod(scale_inv_n(v, m), half, m.q); }, t,
        m, 0, 0);
  }
}

template <int LB>
__global__ __launch_bounds__((1 << LB) / 16) void k_fused_ks_moddown(DevCtx c, const u64 *ksacc, const u64 *tlast, u64 *out, int nl) {
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
  const u64 *src = tlast + cc * N;
  const u64 *ks = ksacc + (cc * nl + j) * N;
  u64 *o = out + (cc * nl + j) * N;
  ntt_fwd_block<LB>(
      lds, [&](int, int i) { return add_mod(reduce64(src[i], m), fix, m.q); },
      [&](int, int i, u64 v) {
        const u64 x = canon4(v, m);
        o[i] = add_mod(mul_shoup(sub_mod(ks[i], x, m.q), inv, inv_s, m.q), o[i], m.q);
      },
      t, m, 0, 0);
}

template <int LB>
static int run_fused(abc_hip_ctx *c, const u64 *a, const u64 *b, u64 *out, int nl, size_t count, size_t chunk) {
  const size_t N = (size_t)1 << LB;
  const size_t per_ct = (size_t)(4 * nl + 2) * N;
  if (ensure_workspace(c, chunk * per_ct * 8)) return 1;
  u64 *c2coef = (u64 *)c->ws, *c2ntt = c2coef + chunk * nl * N;
  u64 *ksacc = c2ntt + chunk * nl * N, *tlast = ksacc + chunk * 2 * nl * N;
  const dim3 block((1 << LB) / 16);
  for (size_t off = 0; off < count; off += chunk) {
    const size_t cc = (count - off < chunk) ? count - off : chunk;
    const size_t ctw = 2 * (size_t)nl * N;
    hipLaunchKernelGGL(k_fused_tensor_intt<LB>, dim3((unsigned)(cc * nl)), block, 0, c->stream, c->dc, a + off * ctw, b + off * ctw,
                       out + off * ctw, c2coef, c2ntt, nl);
    hipLaunchKernelGGL(k_fused_ks_accum<LB>, dim3((unsigned)(cc * (nl + 1))), block, 0, c->stream, c->dc, c2coef, c2ntt,
                       c->d_relin, ksacc, tlast, nl);
    hipLaunchKernelGGL(k_fused_ks_moddown<LB>, dim3((unsigned)(cc * 2 * nl)), block, 0, c->stream, c->dc, ksacc, tlast,
                       out + off * ctw, nl);
    ABC_HIP_CHECK(hipGetLastError());
  }
  return 0;
}

// -1: not applicable (ring too large for an LDS-resident limb) -> caller takes the generic path
int ckks_mul_relin_fused(abc_hip_ctx *c, const u64 *a, const u64 *b, u64 *out, int nl, size_t count) {
  if (c->logn > 14) return -1;
  if (const char *e = std::getenv("ABC_HIP_NO_FUSED"))
    if (e[0] == '1') return -1;
  if (!count) return 0;
  // chunk: scratch of (4L+2) limbs per ciphertext; default keeps scratch + operands of a chunk
  // within ~half of the 256 MiB Infinity Cache
  size_t chunk = 0;
  if (const char *e = std::getenv("ABC_HIP_CHUNK")) chunk = (size_t)std::atol(e);
  if (!chunk) {
    const size_t per_ct_bytes = (size_t)(4 * nl + 2 + 6 * nl) * c->n * 8;
    chunk = ((size_t)128 << 20) / per_ct_bytes;
    if (chunk < 8) chunk = 8;
  }
  if (chunk > count) chunk = count;
  switch (c->logn) {
    case 10: return run_fused<10>(c, a, b, out, nl, count, chunk);
    case 11: return run_fused<11>(c, a, b, out, nl, count, chunk);
    case 12: return run_fused<12>(c, a, b, out, nl, count, chunk);
    case 13: return run_fused<13>(c, a, b, out, nl, count, chunk);
    case 14: return run_fused<14>(c, a, b, out, nl, count, chunk);
    default: return -1;
  }
}

}  // namespace abc
