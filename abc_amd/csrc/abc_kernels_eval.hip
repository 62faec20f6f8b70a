// abc_kernels_eval.hip -- evaluator kernels shared by BFV and CKKS (generic, any N = 2^10..2^16):
// limb-wise add/sub/negate, CKKS dyadic tensor, hybrid key switching with one special prime,
// Galois automorphisms, rescale / modulus drop.
//
// Reference call sites replaced (src/runtime/SealCiphertext.cpp): add :92,:114; sub :98,:118;
// negate :157,:193; relinearize_inplace :105,:123,:160,:197; rotate_rows :55,:60.
// Key switching follows seal::Evaluator::switch_key_inplace: decomposition limb J is reduced modulo
// every key-level prime, transformed, multiplied with key[J] and accumulated; the special-prime limb is
// then divided out with rounding.  All outputs are fully reduced, hence bit-comparable with oracle/.
#include <cstdlib>

#include "abc_context.hpp"

namespace abc {

LimbMap key_limb_map(const abc_hip_ctx *c, int nl) {
  LimbMap m{};
  for (int j = 0; j < nl; j++) m.id[j] = j;
  m.id[nl] = c->K - 1;
  return m;
}

static inline unsigned grid_for(size_t items, int block) {
  size_t g = (items + block - 1) / block;
  const size_t cap = 256 * 8 * 4;  // enough workgroups to fill 256 CUs several times, grid-stride beyond
  return (unsigned)(g < cap ? (g ? g : 1) : cap);
}

// ---- limb-wise add / sub / negate:  data viewed as [polys][nl][N] ----
__global__ __launch_bounds__(256) void k_addsub(DevCtx c, const u64 *a, const u64 *b, u64 *out, int nl, size_t words, int op) {
  const size_t stride = (size_t)gridDim.x * blockDim.x * 2;
  for (size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; w < words; w += stride) {
    const int j = (int)((w >> c.logn) % nl);
    const u64 q = c.mods[j].q;
    u64x2 x = *reinterpret_cast<const u64x2 *>(a + w);
    u64x2 r;
    if (op == 2) {
      r.x = neg_mod(x.x, q);
      r.y = neg_mod(x.y, q);
    } else {
      u64x2 y = *reinterpret_cast<const u64x2 *>(b + w);
      if (op == 0) { r.x = add_mod(x.x, y.x, q); r.y = add_mod(x.y, y.y, q); }
      else { r.x = sub_mod(x.x, y.x, q); r.y = sub_mod(x.y, y.y, q); }
    }
    *reinterpret_cast<u64x2 *>(out + w) = r;
  }
}

int launch_addsub(abc_hip_ctx *c, const u64 *a, const u64 *b, u64 *out, int nl, size_t polys, int op) {
  const size_t words = polys * nl * (size_t)c->n;
  if (!words) return 0;
  hipLaunchKernelGGL(k_addsub, dim3(grid_for(words / 2, 256)), dim3(256), 0, c->stream, c->dc, a, b, out, nl, words, op);
  ABC_HIP_CHECK(hipGetLastError());
  return 0;
}

// ---- CKKS tensor product: (a0,a1) x (b0,b1) -> (a0b0, a0b1+a1b0, a1b1), NTT form ----
__global__ __launch_bounds__(256) void k_ckks_tensor(DevCtx c, const u64 *a, const u64 *b, u64 *out3, int nl, size_t count) {
  const size_t pw = (size_t)nl * c.n;  // words per polynomial
  const size_t items = count * pw;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const size_t ct = it / pw, w = it % pw;
    const Mod m = c.mods[w >> c.logn];
    const u64 *pa = a + ct * 2 * pw + w, *pb = b + ct * 2 * pw + w;
    const u64 a0 = pa[0], a1 = pa[pw], b0 = pb[0], b1 = pb[pw];
    u64 *po = out3 + ct * 3 * pw + w;
    po[0] = mul_mod(a0, b0, m);
    U128 acc = mul_wide(a0, b1);
    mac128(acc, a1, b0);
    po[pw] = barrett_reduce(acc, m);
    po[2 * pw] = mul_mod(a1, b1, m);
  }
}

int launch_ckks_tensor(abc_hip_ctx *c, const u64 *a, const u64 *b, u64 *out3, int nl, size_t count) {
  const size_t items = count * nl * (size_t)c->n;
  if (!items) return 0;
  hipLaunchKernelGGL(k_ckks_tensor, dim3(grid_for(items, 256)), dim3(256), 0, c->stream, c->dc, a, b, out3, nl, count);
  ABC_HIP_CHECK(hipGetLastError());
  return 0;
}

// ---- key switching, generic path ----
// dec[ct][J][I][k] = tcoef[ct][J][k] mod q_{ki(I)}
__global__ __launch_bounds__(256) void k_ks_expand(DevCtx c, const u64 *tcoef, size_t tstride, u64 *dec, int nl, size_t count) {
  const size_t per_ct = (size_t)nl * c.n;
  const size_t items = count * per_ct;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const size_t ct = it / per_ct, r = it % per_ct;
    const int J = (int)(r >> c.logn);
    const size_t k = r & (c.n - 1);
    const u64 v = tcoef[ct * tstride + r];
    u64 *d = dec + ((ct * nl + J) * (size_t)(nl + 1)) * c.n + k;
    for (int I = 0; I <= nl; I++) {
      const int ki = (I == nl) ? c.K - 1 : I;
      d[(size_t)I * c.n] = reduce64(v, c.mods[ki]);
    }
  }
}

// prodD[ct][comp][I<nl][k], prodS[ct][comp][k] = sum_J dec[ct][J][I][k] * key[J][comp][ki][k]
__global__ __launch_bounds__(256) void k_ks_inner(DevCtx c, const u64 *dec, const u64 *key, u64 *prodD, u64 *prodS, int nl,
                                                  size_t count) {
  const size_t per_ct = (size_t)(nl + 1) * c.n;
  const size_t items = count * per_ct;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const size_t ct = it / per_ct, r = it % per_ct;
    const int I = (int)(r >> c.logn);
    const size_t k = r & (c.n - 1);
    const int ki = (I == nl) ? c.K - 1 : I;
    const Mod m = c.mods[ki];
    u64 acc0 = 0, acc1 = 0;
    for (int J = 0; J < nl; J++) {
      const u64 x = dec[((ct * nl + J) * (size_t)(nl + 1) + I) * c.n + k];
      const u64 *kj = key + (((size_t)J * 2) * c.K + ki) * c.n + k;
      acc0 = add_mod(acc0, mul_mod(x, kj[0], m), m.q);
      acc1 = add_mod(acc1, mul_mod(x, kj[(size_t)c.K * c.n], m), m.q);
    }
    if (I == nl) {
      prodS[(ct * 2 + 0) * c.n + k] = acc0;
      prodS[(ct * 2 + 1) * c.n + k] = acc1;
    } else {
      prodD[((ct * 2 + 0) * nl + I) * c.n + k] = acc0;
      prodD[((ct * 2 + 1) * nl + I) * c.n + k] = acc1;
    }
  }
}

// tmod[ct][comp][j][k] = ((prodS + half) mod q_sp) mod q_j  + (q_j - half mod q_j)
__global__ __launch_bounds__(256) void k_ks_tmod(DevCtx c, const u64 *prodS, u64 *tmod, int nl, size_t polys) {
  const size_t items = polys * c.n;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const Mod msp = c.mods[c.K - 1];
  const u64 half = msp.q >> 1;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const size_t p = it >> c.logn, k = it & (c.n - 1);
    const u64 last = add_mod(prodS[it], half, msp.q);
    for (int j = 0; j < nl; j++) {
      const Mod m = c.mods[j];
      const u64 fix = m.q - reduce64(half, m);
      tmod[(p * nl + j) * c.n + k] = add_mod(reduce64(last, m), fix == m.q ? 0 : fix, m.q);
    }
  }
}

// out[ct][comp][j][k] = (prodD - tmod) * q_sp^-1 mod q_j  (+ addend)
__global__ __launch_bounds__(256) void k_ks_finish(DevCtx c, const u64 *prodD, const u64 *tmod, u64 *out, const u64 *addend,
                                                   size_t addend_stride, int add_c1, int nl, size_t count) {
  const size_t pw = (size_t)nl * c.n;
  const size_t items = count * 2 * pw;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const size_t ct = it / (2 * pw), r = it % (2 * pw);
    const int comp = (int)(r / pw);
    const size_t w = r % pw;
    const int j = (int)(w >> c.logn);
    const u64 q = c.mods[j].q;
    u64 v = mul_shoup(sub_mod(prodD[it], tmod[it], q), c.cst->inv_special[j], c.cst->inv_special_s[j], q);
    if (addend && (comp == 0 || add_c1)) v = add_mod(v, addend[ct * addend_stride + r], q);
    out[it] = v;
  }
}

int launch_ks_tmod(abc_hip_ctx *c, const u64 *prodS, u64 *tmod, int nl, size_t polys) {
  hipLaunchKernelGGL(k_ks_tmod, dim3(grid_for(polys * c->n, 256)), dim3(256), 0, c->stream, c->dc, prodS, tmod, nl, polys);
  ABC_HIP_CHECK(hipGetLastError());
  return 0;
}
int launch_ks_finish(abc_hip_ctx *c, const u64 *prodD, const u64 *tmod, u64 *out, const u64 *addend, size_t addend_stride,
                     bool add_c1, int nl, size_t count) {
  hipLaunchKernelGGL(k_ks_finish, dim3(grid_for(count * 2 * nl * (size_t)c->n, 256)), dim3(256), 0, c->stream, c->dc, prodD, tmod,
                     out, addend, addend_stride, add_c1 ? 1 : 0, nl, count);
  ABC_HIP_CHECK(hipGetLastError());
  return 0;
}

// ---- N = 2^15 / 2^16, integer arithmetic (a key prime above 2^50: config 5's 55-bit chain): decomposition and inner product
// in two kernels instead of five.  The generic sequence writes every digit once per key prime (k_ks_expand), carries those
// nl (nl + 1) limbs through the strided pass and the block transforms (two round trips) and reads them again in k_ks_inner:
// five transfers of nl (nl + 1) limbs.  Here:
//   k_iks_pass0   (ct, digit J, key prime I, 256 positions): digit J read as it lies, reduced modulo q_I, strided first
//                 stages -> half-done limb (ct, J, I)                                        [one write]
//   k_iks_special (ct, key prime I, 4096-point block b): per digit J the block stages in LDS, the transform's outputs go
//                 straight from registers into the two inner products with the key (same register layout for every J and
//                 for the inverse block transform that follows), then the inverse block stages of both sums -> prodD /
//                 prodS half-done (the strided last stages, q_sp rounding and the subtraction stay with the generic
//                 kernels)                                                                    [one read, 2 / nl of a write]
// Inner products: canonical operands, one Barrett product per term (a 128-bit accumulator per value would cost 128 VGPRs).
template <int R>
__global__ __launch_bounds__(256) void k_iks_pass0(DevCtx c, const u64 *__restrict__ tcoef, size_t tstride, u64 *__restrict__ dec, int nl,
                                                   u32 ginv /* BFV rotation: elt^-1 mod 2N, the signed permutation folded into the load */) {
  const int G = c.n >> R;
  const int per = G / 256;
  const size_t limb = blockIdx.x / per;  // (ct * nl + J) * (nl + 1) + I
  const int p = (blockIdx.x % per) * 256 + threadIdx.x;
  const int I = (int)(limb % (size_t)(nl + 1));
  const size_t cj = limb / (size_t)(nl + 1);
  const int J = (int)(cj % (size_t)nl);
  const size_t ct = cj / (size_t)nl;
  const int ki = (I == nl) ? c.K - 1 : I;
  const Mod m = c.mods[ki];
  const NttTable t = ntt_table(c, ki);
  const u64 *__restrict__ src = tcoef + ct * tstride + (size_t)J * c.n + p;
  u64 x[1 << R];
  if (ginv) {  // workgroup-uniform
    const u64 *__restrict__ limb = tcoef + ct * tstride + (size_t)J * c.n;
    const u64 qj = c.mods[J].q;
#pragma unroll
    for (int k = 0; k < (1 << R); k++) {
      bool neg;
      const u64 v = limb[galois_coef_src((u32)(k * G + p), ginv, c.logn, neg)];
      x[k] = neg ? neg_mod(v, qj) : v;
    }
  } else {
#pragma unroll
    for (int k = 0; k < (1 << R); k++) x[k] = src[(size_t)k * G];
  }
#pragma unroll
  for (int k = 0; k < (1 << R); k++) x[k] = reduce64(x[k], m);
#pragma unroll
  for (int u = 0; u < R; u++) {
    const int half = 1 << (R - 1 - u);
#pragma unroll
    for (int k = 0; k < (1 << R); k++) {
      if (k & half) continue;
      const u64x2 tp = tw_load(t.tw + (1 << u) + (k >> (R - u)));
      const u64 a = csub(x[k], m.two_q);
      const u64 v = mul_shoup_lazy(x[k | half], tp.x, tp.y, m.q);
      x[k] = a + v;
      x[k | half] = a + m.two_q - v;
    }
  }
  u64 *__restrict__ dst = dec + limb * (size_t)c.n + p;
#pragma unroll
  for (int k = 0; k < (1 << R); k++) dst[(size_t)k * G] = x[k];  // lazy [0, 4q): the block stages are guarded
}

// Shoup quotients of a key-switching key, floor(w 2^64 / q) per word, same layout: with them a term of the inner product is one
// lazy Shoup product of ANY 64-bit transform output (no canonicalisation, no 128-bit product, no Barrett): built on first use,
// dropped with the key's other mirror (drop_key_twins).
__global__ __launch_bounds__(256) void k_key_to_shoup(DevCtx c, const u64 *__restrict__ key, u64 *__restrict__ ks, size_t words) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += stride) {
    const int kp = (int)((i >> c.logn) % (size_t)c.K);
    const u64 q = c.mods[kp].q;
    u64 r = key[i], quo = 0;  // r < q < 2^61: schoolbook division of r 2^64 by q, one quotient bit per step
    for (int bit = 0; bit < 64; bit++) {
      r <<= 1;
      const bool ge = r >= q;
      r -= ge ? q : 0;
      quo = (quo << 1) | (ge ? 1u : 0u);
    }
    ks[i] = quo;
  }
}
static const u64 *key_shoup(abc_hip_ctx *c, const u64 *key) {
  if (c->sw.no_key_twin || !key) return nullptr;
  auto it = c->key_shoups.find(key);
  if (it != c->key_shoups.end()) return it->second;
  if (c->capture_active) return nullptr;  // built by the eager pass that precedes every recording
  u64 *d = nullptr;
  const size_t words = c->key_words();
  if (hipMalloc(&d, words * 8) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  hipLaunchKernelGGL(k_key_to_shoup, dim3(grid_for(words, 256)), dim3(256), 0, c->stream, c->dc, key, d, words);
  c->key_shoups[key] = d;
  return d;
}

// INV_D: the data limbs' sums go back to coefficients too (BFV); CKKS keeps them in NTT form.  SHOUP: `keys` = the key's Shoup
// quotients; sums of lazy products (< 2q each) stay below 2^64 for up to 16 digits of a 59-bit prime, wider primes fold after
// every term.
template <int LB, bool INV_D, bool SHOUP, bool GUARD>
__global__ __launch_bounds__((1 << LB) / 16, 3) void k_iks_special(DevCtx c, const u64 *__restrict__ dec, const u64 *__restrict__ key,
                                                                    const u64 *__restrict__ keys, u64 *__restrict__ prodD,
                                                                    u64 *__restrict__ prodS, int nl, int S0, unsigned cc) {
  __shared__ u64 lds[lds_words(LB)];
  // Every workgroup of a (key prime, block) pair reads the same 2 nl (x 2 with quotients) key blocks: ciphertext index fastest,
  // and -- consecutive workgroup ids go to consecutive XCDs -- the id is remapped so that one XCD's L2 sees the whole run
  // of ciphertexts of a pair (the grid is a multiple of 8: 2^S0 blocks per limb).
  const unsigned per_xcd = gridDim.x >> 3;
  const unsigned wid = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  const size_t ct = wid % cc;
  const unsigned pair = wid / cc;
  const int b = (int)(pair & ((1u << S0) - 1));
  const int I = (int)(pair >> S0);
  const int ki = (I == nl) ? c.K - 1 : I;
  const Mod m = c.mods[ki];
  const NttTable t = ntt_table(c, ki);
  const size_t N = (size_t)c.n, off = (size_t)b << LB;
  const bool fold = m.bits > 59;
  u64 acc0[16], acc1[16];
#pragma unroll
  for (int r = 0; r < 16; r++) acc0[r] = acc1[r] = 0;
#pragma nounroll
  for (int J = 0; J < nl; J++) {
    const u64 *__restrict__ in = dec + ((ct * nl + J) * (size_t)(nl + 1) + I) * N + off;
    const size_t kw = (((size_t)J * 2) * c.K + ki) * N + off, kw1 = kw + (size_t)c.K * N;
    ntt_fwd_block<LB, GUARD>(
        lds, [&](int, int i) { return in[i]; },
        [&](int r, int i, u64 v) {
          if (SHOUP) {
            acc0[r] += mul_shoup_lazy(v, key[kw + i], keys[kw + i], m.q);
            acc1[r] += mul_shoup_lazy(v, key[kw1 + i], keys[kw1 + i], m.q);
            if (fold) {
              acc0[r] = csub(acc0[r], m.two_q);
              acc1[r] = csub(acc1[r], m.two_q);
            }
          } else {
            const u64 x = canon_fwd<GUARD>(v, m);
            acc0[r] = add_mod(acc0[r], mul_mod(x, key[kw + i], m), m.q);
            acc1[r] = add_mod(acc1[r], mul_mod(x, key[kw1 + i], m), m.q);
          }
        },
        t, m, S0, b);
    block_sync_lds();  // the next transform's first pass rewrites words other wavefronts have just read
  }
  if (SHOUP) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      acc0[r] = reduce64(acc0[r], m);
      acc1[r] = reduce64(acc1[r], m);
    }
  }
  u64 *__restrict__ d0 = (I == nl ? prodS + (ct * 2 + 0) * N : prodD + ((ct * 2 + 0) * nl + I) * N) + off;
  u64 *__restrict__ d1 = (I == nl ? prodS + (ct * 2 + 1) * N : prodD + ((ct * 2 + 1) * nl + I) * N) + off;
  if (!INV_D && I != nl) {
    using P = PassIdx<LB, LB - 2, 2>;  // the forward transform's final register layout
    int hi[P::NG], lo[P::NG];
    P::groups((int)threadIdx.x, hi, lo);
#pragma unroll
    for (int g = 0; g < P::NG; g++)
#pragma unroll
      for (int k = 0; k < 4; k++) {
        d0[P::elem(hi[g], lo[g], k)] = acc0[g * 4 + k];
        d1[P::elem(hi[g], lo[g], k)] = acc1[g * 4 + k];
      }
    return;
  }
  ntt_inv_block<LB>(lds, [&](int r, int) { return acc0[r]; }, [&](int, int i, u64 v) { d0[i] = v; }, t, m, S0, b);
  block_sync_lds();
  ntt_inv_block<LB>(lds, [&](int r, int) { return acc1[r]; }, [&](int, int i, u64 v) { d1[i] = v; }, t, m, S0, b);
}

// BFV: everything behind k_iks_special in one kernel -- (ct, component, 256 positions): the strided last stages of the special
// limb's inverse transform, N^-1, + q_sp/2 -> t (2^R values in registers); then per data prime the same stages on its limb,
// N^-1, minus (t mod q_j + the rounding fix), times q_sp^-1, plus the addend -> coefficient form.  Replaces two strided passes
// (in place: a read and a write of 2 (nl + 1) limbs), k_ks_tmod (2 nl limbs written, read back) and k_ks_finish.
template <int R>
__global__ __launch_bounds__(256) void k_iks_finish(DevCtx c, const u64 *__restrict__ prodD, const u64 *__restrict__ prodS,
                                                    const u64 *addend, size_t addend_stride, int add_c1, u64 *out /* may be the addend */,
                                                    int nl, u32 ginv /* BFV rotation: gather the addend (then out is not the addend) */) {
  const int G = c.n >> R;
  const int per = G / 256;
  const size_t cc = blockIdx.x / per;  // ct * 2 + comp
  const int p = (blockIdx.x % per) * 256 + threadIdx.x;
  const size_t ct = cc >> 1;
  const int comp = (int)(cc & 1);
  const size_t N = (size_t)c.n;
  auto inverse_stages = [&](u64(&x)[1 << R], const Mod &m, const NttTable &t) {
#pragma unroll
    for (int u = R - 1; u >= 0; u--) {
      const int half = 1 << (R - 1 - u);
#pragma unroll
      for (int k = 0; k < (1 << R); k++) {
        if (k & half) continue;
        const u64x2 tp = tw_load(t.itw + (1 << u) + (k >> (R - u)));
        const u64 a = x[k], b2 = x[k | half];
        x[k] = csub(a + b2, m.two_q);
        x[k | half] = mul_shoup_lazy(a + m.two_q - b2, tp.x, tp.y, m.q);
      }
    }
  };
  u64 t[1 << R];
  {
    const Mod ms = c.mods[c.K - 1];
    const u64 *__restrict__ src = prodS + cc * N + p;
#pragma unroll
    for (int k = 0; k < (1 << R); k++) t[k] = src[(size_t)k * G];
    inverse_stages(t, ms, ntt_table(c, c.K - 1));
    const u64 half = ms.q >> 1;
#pragma unroll
    for (int k = 0; k < (1 << R); k++) t[k] = add_mod(scale_inv_n(t[k], ms), half, ms.q);
  }
  const u64 half = c.mods[c.K - 1].q >> 1;
  const bool add = addend && (comp == 0 || add_c1);
#pragma nounroll
  for (int j = 0; j < nl; j++) {
    const Mod m = c.mods[j];
    const u64 *__restrict__ src = prodD + (cc * nl + j) * N + p;
    u64 x[1 << R];
#pragma unroll
    for (int k = 0; k < (1 << R); k++) x[k] = src[(size_t)k * G];
    inverse_stages(x, m, ntt_table(c, j));
    const u64 fix0 = m.q - reduce64(half, m), fix = fix0 == m.q ? 0 : fix0;
    const u64 inv = c.cst->inv_special[j], inv_s = c.cst->inv_special_s[j];
    u64 *o = out + (cc * nl + j) * N + p;
    const u64 *cin = add ? addend + ct * addend_stride + ((size_t)comp * nl + j) * N + p : nullptr;
#pragma unroll
    for (int k = 0; k < (1 << R); k++) {
      const u64 tm = add_mod(reduce64(t[k], m), fix, m.q);
      u64 v = mul_shoup(sub_mod(scale_inv_n(x[k], m), tm, m.q), inv, inv_s, m.q);
      if (add) {
        if (ginv) {  // workgroup-uniform
          bool neg;
          const u64 a = (cin - p)[galois_coef_src((u32)(k * G + p), ginv, c.logn, neg)];
          v = add_mod(v, neg ? neg_mod(a, m.q) : a, m.q);
        } else {
          v = add_mod(v, cin[(size_t)k * G], m.q);
        }
      }
      o[(size_t)k * G] = v;
    }
  }
}
static int iks_finish(abc_hip_ctx *c, const u64 *prodD, const u64 *prodS, const u64 *addend, size_t addend_stride, bool add_c1, u64 *out,
                      int nl, size_t cc, u32 ginv) {
  const int S0 = c->logn - big_block_log();
  const int G = c->n >> S0;
  const dim3 grid((unsigned)(cc * 2 * (G / 256)));
  if (S0 == 3)
    hipLaunchKernelGGL(k_iks_finish<3>, grid, dim3(256), 0, c->stream, c->dc, prodD, prodS, addend, addend_stride, add_c1 ? 1 : 0, out, nl, ginv);
  else
    hipLaunchKernelGGL(k_iks_finish<4>, grid, dim3(256), 0, c->stream, c->dc, prodD, prodS, addend, addend_stride, add_c1 ? 1 : 0, out, nl, ginv);
  ABC_HIP_CHECK(hipGetLastError());
  return 0;
}

// 1: error; -1: not applicable (the caller takes the generic kernels); 0: prodS holds the special limb's sums half-way back to
// coefficients (strided stages left), prodD the data limbs' -- likewise for BFV, in NTT form for CKKS
// BFV with the permutation of a rotation folded in (rotate_fused): the integer sequence must be the one that runs
bool iks_bfv_applies(const abc_hip_ctx *c, int nl) {
  if (c->scheme != 1 || (c->logn != 15 && c->logn != 16) || c->sw.no_iks || big_block_log() != 12 || nl < 1) return false;
  bool fp = c->use_fp;
  for (int j = 0; j < nl; j++) fp = fp && fp_ok(c->h_mods[j].bits);
  fp = fp && fp_ok(c->h_mods[c->K - 1].bits);
  return !fp;  // an all-fp64 decomposition takes launch_ks_expand_ntt_fp + the generic kernels
}
static int iks_front(abc_hip_ctx *c, const u64 *tc, size_t tcs, const u64 *key, u64 *dec, u64 *prodD, u64 *prodS, int nl, size_t cc, u32 ginv) {
  if ((c->logn != 15 && c->logn != 16) || c->sw.no_iks) return -1;
  const int S0 = c->logn - big_block_log();
  if (big_block_log() != 12) return -1;
  const size_t limbs = cc * nl * (nl + 1);
  const int G = c->n >> S0;
  const dim3 g0((unsigned)(limbs * (G / 256)));
  if (S0 == 3) hipLaunchKernelGGL(k_iks_pass0<3>, g0, dim3(256), 0, c->stream, c->dc, tc, tcs, dec, nl, ginv);
  else hipLaunchKernelGGL(k_iks_pass0<4>, g0, dim3(256), 0, c->stream, c->dc, tc, tcs, dec, nl, ginv);
  ABC_HIP_CHECK(hipGetLastError());
  const dim3 g1((unsigned)((cc * (nl + 1)) << S0));
  const u64 *keys = key_shoup(c, key);
  const bool ckks = c->scheme == 2;
  // unguarded butterflies where every key prime leaves the room: [0, 4q) out of the strided pass, + 4q per block stage = 52q < 2^64
  bool guard = false;
  for (int j = 0; j < c->K; j++) guard = guard || !unguarded_ok(c->h_mods[j].bits);
#define ABC_IKS(INV_D, SHOUP)                                                                                                        \
  do {                                                                                                                               \
    if (guard)                                                                                                                       \
      hipLaunchKernelGGL((k_iks_special<12, INV_D, SHOUP, true>), g1, dim3(256), 0, c->stream, c->dc, dec, key, keys, prodD, prodS, \
                         nl, S0, (unsigned)cc);                                                                                      \
    else                                                                                                                             \
      hipLaunchKernelGGL((k_iks_special<12, INV_D, SHOUP, false>), g1, dim3(256), 0, c->stream, c->dc, dec, key, keys, prodD, prodS, \
                         nl, S0, (unsigned)cc);                                                                                      \
  } while (0)
  if (keys) { if (ckks) ABC_IKS(false, true); else ABC_IKS(true, true); }
  else { if (ckks) ABC_IKS(false, false); else ABC_IKS(true, false); }
#undef ABC_IKS
  ABC_HIP_CHECK(hipGetLastError());
  return 0;
}

int keyswitch_generic(abc_hip_ctx *c, const u64 *target, size_t target_stride, const u64 *key, u64 *out2, int nl, size_t count,
                      const u64 *addend, size_t addend_stride, bool add_c1, u32 ginv) {
  if (!count) return 0;
  if (ginv && !iks_bfv_applies(c, nl)) { set_error("keyswitch_generic: a folded permutation needs the fused integer sequence"); return 1; }
  const bool ckks = (c->scheme == 2);
  const size_t N = (size_t)c->n;
  // workspace per ciphertext (words): tcoef nl + dec nl(nl+1) + prodD 2nl + prodS 2 + tmod 2nl
  const size_t per_ct = ((size_t)nl + (size_t)nl * (nl + 1) + 2 * nl + 2 + 2 * nl) * N;
  const size_t budget_words = ((size_t)(c->logn > 14 ? 4 : 1) << 30) / 8;  // <= 1 GiB of scratch per chunk (big rings: 4 GiB)
  size_t chunk = budget_words / per_ct;
  if (chunk < 1) chunk = 1;
  if (chunk > count) chunk = count;
  else if (count % chunk && count / chunk < 8) chunk = (count + count / chunk) / (count / chunk + 1);  // even chunks, no runt
  if (ensure_workspace(c, chunk * per_ct * 8)) return 1;
  u64 *tcoef = (u64 *)c->ws;
  u64 *dec = tcoef + chunk * nl * N;
  u64 *prodD = dec + chunk * nl * (nl + 1) * N;
  u64 *prodS = prodD + chunk * 2 * nl * N;
  u64 *tmod = prodS + chunk * 2 * N;
  const LimbMap dmap = key_limb_map(c, nl);
  LimbMap smap{};
  smap.id[0] = c->K - 1;
  for (size_t off = 0; off < count; off += chunk) {
    const size_t cc = (count - off < chunk) ? count - off : chunk;
    const u64 *tg = target + off * target_stride;
    const u64 *tc = tg;
    size_t tcs = target_stride;
    if (ckks) {  // CKKS targets are in NTT form: back to coefficients first
      ABC_HIP_CHECK(hipMemcpy2DAsync(tcoef, nl * N * 8, tg, target_stride * 8, nl * N * 8, cc, hipMemcpyDeviceToDevice,
                                     c->stream));
      if (launch_ntt_inv(c, tcoef, dmap, nl, cc * nl)) return 1;
      tc = tcoef;
      tcs = nl * N;
    }
    const int fused_expand = launch_ks_expand_ntt_fp(c, tc, tcs, dec, dmap, nl, cc);
    if (fused_expand > 0) return 1;
    const int iks = fused_expand < 0 ? iks_front(c, tc, tcs, key, dec, prodD, prodS, nl, cc, ginv) : -1;
    if (iks > 0) return 1;
    if (iks == 0 && !ckks) {  // inner products done, block stages of the inverse transforms too: one kernel does the rest
      if (iks_finish(c, prodD, prodS, addend ? addend + off * addend_stride : nullptr, addend_stride, add_c1, out2 + off * 2 * nl * N, nl, cc, ginv))
        return 1;
      continue;
    }
    if (iks == 0) {  // CKKS: the special limb's strided stages (integers, like its block stages: k_iks_special), then as ever
      if (launch_ntt_inv_strided_part(c, prodS, smap, 1, cc * 2, true)) return 1;
      if (launch_ks_tmod(c, prodS, tmod, nl, cc * 2)) return 1;
      if (launch_ntt_fwd(c, tmod, dmap, nl, cc * 2 * nl)) return 1;
    } else {
      if (fused_expand < 0) {
        hipLaunchKernelGGL(k_ks_expand, dim3(grid_for(cc * nl * N, 256)), dim3(256), 0, c->stream, c->dc, tc, tcs, dec, nl, cc);
        ABC_HIP_CHECK(hipGetLastError());
        if (launch_ntt_fwd(c, dec, dmap, nl + 1, cc * nl * (nl + 1))) return 1;
      }
      hipLaunchKernelGGL(k_ks_inner, dim3(grid_for(cc * (nl + 1) * N, 256)), dim3(256), 0, c->stream, c->dc, dec, key, prodD,
                         prodS, nl, cc);
      ABC_HIP_CHECK(hipGetLastError());
      if (launch_ntt_inv(c, prodS, smap, 1, cc * 2)) return 1;
      if (launch_ks_tmod(c, prodS, tmod, nl, cc * 2)) return 1;
      if (ckks) {
        if (launch_ntt_fwd(c, tmod, dmap, nl, cc * 2 * nl)) return 1;
      } else {
        if (launch_ntt_inv(c, prodD, dmap, nl, cc * 2 * nl)) return 1;
      }
    }
    if (launch_ks_finish(c, prodD, tmod, out2 + off * 2 * nl * N, addend ? addend + off * addend_stride : nullptr, addend_stride,
                         add_c1, nl, cc))
      return 1;
  }
  return 0;
}

// ---- Galois automorphism x -> x^elt on [polys][nl][N] ----
__global__ __launch_bounds__(256) void k_galois(DevCtx c, const u64 *in, u64 *out, int nl, size_t polys, u32 elt, int ntt_form) {
  const size_t items = polys * nl * (size_t)c.n;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const size_t limb = it >> c.logn;
    const u32 i = (u32)(it & (c.n - 1));
    const u64 *src = in + limb * c.n;
    u64 *dst = out + limb * c.n;
    if (ntt_form) {
      // slot i holds the evaluation at psi^(2*bitrev(i)+1); gather from the slot holding exponent*elt
      const u32 rev = bitrev32(i + (u32)c.n, c.logn + 1);
      const u64 idx = (((u64)elt * rev) >> 1) & (u64)(c.n - 1);
      dst[i] = src[bitrev32((u32)idx, c.logn)];
    } else {
      const u64 raw = (u64)i * elt;
      const u32 idx = (u32)(raw & (u64)(c.n - 1));
      u64 v = src[i];
      if ((raw >> c.logn) & 1) v = neg_mod(v, c.mods[limb % nl].q);
      dst[idx] = v;
    }
  }
}

int launch_galois(abc_hip_ctx *c, const u64 *in, u64 *out, int nl, size_t polys, uint32_t elt, bool ntt_form) {
  const size_t items = polys * nl * (size_t)c->n;
  if (!items) return 0;
  hipLaunchKernelGGL(k_galois, dim3(grid_for(items, 256)), dim3(256), 0, c->stream, c->dc, in, out, nl, polys, elt,
                     ntt_form ? 1 : 0);
  ABC_HIP_CHECK(hipGetLastError());
  return 0;
}

// ---- CKKS rescale: drop limb nl-1 with rounding (divide_and_round_q_last_ntt) ----
__global__ __launch_bounds__(256) void k_gather_last(DevCtx c, const u64 *in, u64 *last, int nl, size_t polys) {
  const size_t items = polys * c.n;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const size_t p = it >> c.logn, k = it & (c.n - 1);
    last[it] = in[(p * nl + (nl - 1)) * c.n + k];
  }
}
__global__ __launch_bounds__(256) void k_rescale_tmod(DevCtx c, const u64 *last, u64 *tmod, int nl, size_t polys) {
  const size_t items = polys * c.n;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const Mod ml = c.mods[nl - 1];
  const u64 half = ml.q >> 1;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const size_t p = it >> c.logn, k = it & (c.n - 1);
    const u64 v = add_mod(last[it], half, ml.q);
    for (int j = 0; j < nl - 1; j++) {
      const Mod m = c.mods[j];
      const u64 hm = reduce64(half, m);
      tmod[(p * (nl - 1) + j) * c.n + k] = add_mod(reduce64(v, m), hm ? m.q - hm : 0, m.q);
    }
  }
}
__global__ __launch_bounds__(256) void k_rescale_finish(DevCtx c, const u64 *in, const u64 *tmod, u64 *out, int nl, size_t polys) {
  const int nlo = nl - 1;
  const size_t items = polys * nlo * (size_t)c.n;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const size_t limb = it >> c.logn, k = it & (c.n - 1);
    const size_t p = limb / nlo;
    const int j = (int)(limb % nlo);
    const u64 q = c.mods[j].q;
    const u64 x = in[(p * nl + j) * c.n + k];
    out[it] = mul_shoup(sub_mod(x, tmod[it], q), c.cst->inv_qlast[nl - 1][j], c.cst->inv_qlast_s[nl - 1][j], q);
  }
}

// ---- the same for rings that fit LDS and primes below 2^50: two kernels instead of five ----
// R1 (poly): inverse transform of the last limb, read in place, + q_last/2 -> scratch.
// R2 (poly, j < nl-1): scratch + rounding fix as operand of a forward transform modulo q_j in LDS (no reduction: the
//     fp64 bound absorbs a 50-bit operand), then (in_j - transform) * q_last^-1 -> out_j.  The nl-1 workgroups that read one
//     scratch polynomial are 8 apart in blockIdx (one XCD, one L2).  7 + 2 limb transfers per polynomial instead of 23.
template <int LB>
__global__ __launch_bounds__((1 << LB) / 16) void k_rescale_intt_fp(DevCtx c, const u64 *__restrict__ in, u64 *__restrict__ last,
                                                                    int nl) {
  __shared__ double lds[lds_words(LB)];
  const size_t N = (size_t)1 << LB;
  const Mod m = mod_at(c, nl - 1);
  const FpTable t = fp_table(c, nl - 1);
  const double half = (double)(m.q >> 1);
  const u64 *__restrict__ src = in + ((size_t)blockIdx.x * nl + (nl - 1)) * N;
  u64 *__restrict__ dst = last + (size_t)blockIdx.x * N;
  ntt_inv_block_a<LB, FpArith>(
      lds, [&](int, int i) { return fp_from_u64(src[i]); },
      [&](int, int i, double v) { dst[i] = fp_to_canon(fp_mul_lazy(v, m.inv_n_c, m.inv_n_cq, m.qd) + half, m.qd, m.qinv); }, t, m, 0,
      0);
}
template <int LB>
__global__ __launch_bounds__((1 << LB) / 16) void k_rescale_ntt_fp(DevCtx c, const u64 *__restrict__ in, const u64 *__restrict__ last,
                                                                   u64 *__restrict__ out, int nl, int npoly) {
  __shared__ double lds[lds_words(LB)];
  const int nlo = nl - 1;
  const unsigned per = 8u * (unsigned)nlo;
  const unsigned grp = blockIdx.x / per, rem = blockIdx.x % per;
  const unsigned left = (unsigned)npoly - grp * 8u, gsz = left < 8u ? left : 8u;  // the last group may be ragged
  const int j = (int)(rem / gsz);
  const size_t p = (size_t)grp * 8 + rem % gsz;
  const size_t N = (size_t)1 << LB;
  const Mod m = mod_at(c, j);
  const FpTable t = fp_table(c, j);
  const u64 half = c.mods[nl - 1].q >> 1;
  const u64 hm = reduce64(half, m);
  const double fix = hm ? (double)(m.q - hm) : 0.0;
  const double inv = c.cst->inv_qlast_c[nl - 1][j], inv_q = c.cst->inv_qlast_cq[nl - 1][j];
  const u64 *__restrict__ src = last + p * N;
  const u64 *__restrict__ x = in + (p * nl + j) * N;
  u64 *__restrict__ o = out + (p * nlo + j) * N;
  ntt_fwd_block_a<LB, FpArith>(
      lds, [&](int, int i) { return fp_from_u64(src[i]) + fix; },
      [&](int, int i, double v) { o[i] = fp_to_canon(fp_mul_lazy(fp_from_u64(x[i]) - v, inv, inv_q, m.qd), m.qd, m.qinv); }, t, m, 0,
      0);
}
template <int LB>
static int launch_rescale_fp(abc_hip_ctx *c, const u64 *in, u64 *out, int nl, size_t polys) {
  const size_t N = (size_t)1 << LB;
  if (ensure_workspace(c, polys * N * 8)) return 1;
  u64 *last = (u64 *)c->ws;
  const dim3 block((1 << LB) / 16);
  hipLaunchKernelGGL(k_rescale_intt_fp<LB>, dim3((unsigned)polys), block, 0, c->stream, c->dc, in, last, nl);
  hipLaunchKernelGGL(k_rescale_ntt_fp<LB>, dim3((unsigned)(polys * (nl - 1))), block, 0, c->stream, c->dc, in, last, out, nl,
                     (int)polys);
  ABC_HIP_CHECK(hipGetLastError());
  return 0;
}

// The same two kernels with the arithmetic chosen per prime (chains that contain a prime above 2^50): the inverse transform runs in
// the arithmetic of the dropped prime, each forward transform in that of its own prime j (bit j of fpmask: below 2^50).
template <int LB>
__global__ __launch_bounds__((1 << LB) / 16) void k_rescale_intt_mixed(DevCtx c, const u64 *__restrict__ in, u64 *__restrict__ last, int nl,
                                                                       int fp_last) {
  __shared__ u64 lds_raw[lds_words(LB)];
  const size_t N = (size_t)1 << LB;
  const u64 *__restrict__ src = in + ((size_t)blockIdx.x * nl + (nl - 1)) * N;
  u64 *__restrict__ dst = last + (size_t)blockIdx.x * N;
  if (fp_last) {
    const Mod m = mod_at(c, nl - 1);
    const FpTable t = fp_table(c, nl - 1);
    const double half = (double)(m.q >> 1);
    ntt_inv_block_a<LB, FpArith>(
        reinterpret_cast<double *>(lds_raw), [&](int, int i) { return fp_from_u64(src[i]); },
        [&](int, int i, double v) { dst[i] = fp_to_canon(fp_mul_lazy(v, m.inv_n_c, m.inv_n_cq, m.qd) + half, m.qd, m.qinv); }, t, m, 0, 0);
  } else {
    const Mod m = c.mods[nl - 1];
    const NttTable t = ntt_table(c, nl - 1);
    const u64 half = m.q >> 1;
    ntt_inv_block<LB>(
        lds_raw, [&](int, int i) { return src[i]; }, [&](int, int i, u64 v) { dst[i] = add_mod(scale_inv_n(v, m), half, m.q); }, t, m, 0, 0);
  }
}
template <int LB>
__global__ __launch_bounds__((1 << LB) / 16) void k_rescale_ntt_mixed(DevCtx c, const u64 *__restrict__ in, const u64 *__restrict__ last,
                                                                      u64 *__restrict__ out, int nl, int npoly, u32 fpmask) {
  __shared__ u64 lds_raw[lds_words(LB)];
  const int nlo = nl - 1;
  const unsigned per = 8u * (unsigned)nlo;
  const unsigned grp = blockIdx.x / per, rem = blockIdx.x % per;
  const unsigned left = (unsigned)npoly - grp * 8u, gsz = left < 8u ? left : 8u;  // the last group may be ragged
  const int j = (int)(rem / gsz);
  const size_t p = (size_t)grp * 8 + rem % gsz;
  const size_t N = (size_t)1 << LB;
  const Mod mi = c.mods[j];
  const u64 half = c.mods[nl - 1].q >> 1;
  const u64 hm = reduce64(half, mi);
  const u64 fixu = hm ? mi.q - hm : 0;
  const u64 *__restrict__ src = last + p * N;
  const u64 *__restrict__ x = in + (p * nl + j) * N;
  u64 *__restrict__ o = out + (p * nlo + j) * N;
  if ((fpmask >> j) & 1u) {  // workgroup-uniform
    const Mod m = mod_at(c, j);
    const FpTable t = fp_table(c, j);
    const double inv = c.cst->inv_qlast_c[nl - 1][j], inv_q = c.cst->inv_qlast_cq[nl - 1][j];
    // the dropped prime may be wider than 2^52: reduce modulo q_j as an integer first (canonical, + fix, below 2 q_j)
    ntt_fwd_block_a<LB, FpArith>(
        reinterpret_cast<double *>(lds_raw), [&](int, int i) { return fp_from_u64(add_mod(reduce64(src[i], mi), fixu, mi.q)); },
        [&](int, int i, double v) { o[i] = fp_to_canon(fp_mul_lazy(fp_from_u64(x[i]) - v, inv, inv_q, m.qd), m.qd, m.qinv); }, t, m, 0, 0);
  } else {
    const NttTable t = ntt_table(c, j);
    const u64 inv = c.cst->inv_qlast[nl - 1][j], inv_s = c.cst->inv_qlast_s[nl - 1][j];
    ntt_fwd_block<LB, true>(
        lds_raw, [&](int, int i) { return add_mod(reduce64(src[i], mi), fixu, mi.q); },
        [&](int, int i, u64 v) { o[i] = mul_shoup(sub_mod(x[i], canon_fwd<true>(v, mi), mi.q), inv, inv_s, mi.q); }, t, mi, 0, 0);
  }
}
template <int LB>
static int launch_rescale_mixed(abc_hip_ctx *c, const u64 *in, u64 *out, int nl, size_t polys) {
  const size_t N = (size_t)1 << LB;
  if (ensure_workspace(c, polys * N * 8)) return 1;
  u64 *last = (u64 *)c->ws;
  u32 fpmask = 0;
  if (c->use_fp && !c->sw.no_mixed)
    for (int j = 0; j < nl; j++)
      if (fp_ok(c->h_mods[j].bits)) fpmask |= 1u << j;
  const dim3 block((1 << LB) / 16);
  hipLaunchKernelGGL(k_rescale_intt_mixed<LB>, dim3((unsigned)polys), block, 0, c->stream, c->dc, in, last, nl, (int)((fpmask >> (nl - 1)) & 1u));
  hipLaunchKernelGGL(k_rescale_ntt_mixed<LB>, dim3((unsigned)(polys * (nl - 1))), block, 0, c->stream, c->dc, in, last, out, nl, (int)polys,
                     fpmask);
  ABC_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_rescale(abc_hip_ctx *c, const u64 *in, u64 *out, int size, int nl, size_t count) {
  if (nl < 2) { set_error("rescale: no limb left to drop"); return 1; }
  const size_t N = (size_t)c->n, polys = count * size;
  if (!polys) return 0;
  bool fp = c->use_fp && c->logn <= 14 && in != out && !c->sw.no_fused;
  for (int j = 0; j < nl; j++) fp = fp && fp_ok(c->h_mods[j].bits);
  if (!fp && c->logn <= 14 && in != out && !c->sw.no_fused && !c->sw.no_isplit) switch (c->logn) {  // a prime above 2^50 in the chain
      case 10: return launch_rescale_mixed<10>(c, in, out, nl, polys);
      case 11: return launch_rescale_mixed<11>(c, in, out, nl, polys);
      case 12: return launch_rescale_mixed<12>(c, in, out, nl, polys);
      case 13: return launch_rescale_mixed<13>(c, in, out, nl, polys);
      case 14: return launch_rescale_mixed<14>(c, in, out, nl, polys);
      default: break;
    }
  if (fp) switch (c->logn) {
      case 10: return launch_rescale_fp<10>(c, in, out, nl, polys);
      case 11: return launch_rescale_fp<11>(c, in, out, nl, polys);
      case 12: return launch_rescale_fp<12>(c, in, out, nl, polys);
      case 13: return launch_rescale_fp<13>(c, in, out, nl, polys);
      case 14: return launch_rescale_fp<14>(c, in, out, nl, polys);
      default: break;
    }
  if (ensure_workspace(c, (polys * N + polys * (nl - 1) * N) * 8)) return 1;
  u64 *last = (u64 *)c->ws, *tmod = last + polys * N;
  hipLaunchKernelGGL(k_gather_last, dim3(grid_for(polys * N, 256)), dim3(256), 0, c->stream, c->dc, in, last, nl, polys);
  ABC_HIP_CHECK(hipGetLastError());
  LimbMap lmap{};
  lmap.id[0] = nl - 1;
  if (launch_ntt_inv(c, last, lmap, 1, polys)) return 1;
  hipLaunchKernelGGL(k_rescale_tmod, dim3(grid_for(polys * N, 256)), dim3(256), 0, c->stream, c->dc, last, tmod, nl, polys);
  ABC_HIP_CHECK(hipGetLastError());
  if (launch_ntt_fwd(c, tmod, key_limb_map(c, nl - 1), nl - 1, polys * (nl - 1))) return 1;
  hipLaunchKernelGGL(k_rescale_finish, dim3(grid_for(polys * (nl - 1) * N, 256)), dim3(256), 0, c->stream, c->dc, in, tmod, out,
                     nl, polys);
  ABC_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_drop_last(abc_hip_ctx *c, const u64 *in, u64 *out, int size, int nl, size_t count) {
  if (nl < 2) { set_error("mod_switch: no limb left to drop"); return 1; }
  const size_t N = (size_t)c->n, polys = count * size;
  if (!polys) return 0;
  ABC_HIP_CHECK(hipMemcpy2DAsync(out, (nl - 1) * N * 8, in, nl * N * 8, (nl - 1) * N * 8, polys, hipMemcpyDeviceToDevice,
                                 c->stream));
  return 0;
}

// ---- CKKS plaintext ops (NTT-form plaintext [nl][N]) ----
__global__ __launch_bounds__(256) void k_ckks_plain(DevCtx c, const u64 *ct, const u64 *plain, size_t plain_stride, u64 *out,
                                                    int size, int nl, size_t count, int op) {
  const size_t pw = (size_t)nl * c.n;
  const size_t items = count * size * pw;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const size_t ctp = it / pw, w = it % pw;
    const size_t ci = ctp / size;
    const int p = (int)(ctp % size);
    const Mod m = c.mods[w >> c.logn];
    const u64 pv = plain[ci * plain_stride + w];
    u64 v = ct[it];
    if (op == 0) v = mul_mod(v, pv, m);
    else if (p == 0) v = (op == 1) ? add_mod(v, pv, m.q) : sub_mod(v, pv, m.q);
    out[it] = v;
  }
}
int ckks_multiply_plain(abc_hip_ctx *c, const u64 *ct, const u64 *plain, size_t plain_stride, u64 *out, int size, int nl,
                        size_t count) {
  const size_t items = count * size * nl * (size_t)c->n;
  if (!items) return 0;
  hipLaunchKernelGGL(k_ckks_plain, dim3(grid_for(items, 256)), dim3(256), 0, c->stream, c->dc, ct, plain, plain_stride, out, size,
                     nl, count, 0);
  ABC_HIP_CHECK(hipGetLastError());
  return 0;
}
int ckks_add_plain(abc_hip_ctx *c, const u64 *ct, const u64 *plain, size_t plain_stride, u64 *out, int size, int nl, size_t count,
                   int sub) {
  const size_t items = count * size * nl * (size_t)c->n;
  if (!items) return 0;
  hipLaunchKernelGGL(k_ckks_plain, dim3(grid_for(items, 256)), dim3(256), 0, c->stream, c->dc, ct, plain, plain_stride, out, size,
                     nl, count, sub ? 2 : 1);
  ABC_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace abc
