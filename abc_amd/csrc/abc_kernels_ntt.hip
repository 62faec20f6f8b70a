// abc_kernels_ntt.hip -- stand-alone batched NTT / INTT launches (one workgroup per limb block).
//
// Replaces the ntt_negacyclic_harvey / inverse_ntt_negacyclic_harvey calls inside the seal::Evaluator,
// BatchEncoder and Decryptor routines the reference invokes (src/runtime/SealCiphertext.cpp:104-105,
// 122-123,159,196; src/runtime/SealCiphertextFactory.cpp:130,150-151).  N <= 2^14: one LDS-resident
// pass per limb.  N = 2^15, 2^16: a strided register pass through HBM does the first (forward) / last
// (inverse) logN-12 stages, then 4096-point LDS blocks do the rest -- no transpose is needed because the
// Cooley-Tukey bit-reversed-output ordering keeps later stages block-local.
#include <cstdlib>

#include "abc_context.hpp"

namespace abc {

template <int LB, bool GUARD>
__global__ __launch_bounds__((1 << LB) / 16) void k_ntt_fwd(DevCtx c, u64 *data, const u64 *src, const u64 *src2, size_t split, LimbMap map, int nl, int S0) {
  __shared__ u64 lds[lds_words(LB)];
  const size_t limb = blockIdx.x >> S0;
  const int b = blockIdx.x & ((1 << S0) - 1);
  const int mid = map.id[limb % nl];
  const Mod m = c.mods[mid];
  const NttTable t = ntt_table(c, mid);
  u64 *base = data + limb * (size_t)c.n + ((size_t)b << LB);
  // limbs [0, split) are read from src, the rest from src2 (two operands, one launch); in place: src == data
  const u64 *in = (limb < split ? src + limb * (size_t)c.n : src2 + (limb - split) * (size_t)c.n) + ((size_t)b << LB);
  ntt_fwd_block<LB, GUARD>(
      lds, [&](int, int i) { return in[i]; }, [&](int, int i, u64 v) { base[i] = canon_fwd<GUARD>(v, m); }, t, m, S0, b);
}

template <int LB>
__global__ __launch_bounds__((1 << LB) / 16) void k_ntt_inv(DevCtx c, u64 *data, LimbMap map, int nl, int S0) {
  __shared__ u64 lds[lds_words(LB)];
  const size_t limb = blockIdx.x >> S0;
  const int b = blockIdx.x & ((1 << S0) - 1);
  const int mid = map.id[limb % nl];
  const Mod m = c.mods[mid];
  const NttTable t = ntt_table(c, mid);
  u64 *base = data + limb * (size_t)c.n + ((size_t)b << LB);
  const bool whole = (S0 == 0);
  ntt_inv_block<LB>(
      lds, [&](int, int i) { return base[i]; }, [&](int, int i, u64 v) { base[i] = whole ? scale_inv_n(v, m) : v; }, t, m,
      S0, b);
}

// fp64 variants (every prime of the launch below 2^50, whole transform in one block): same memory format, the
// conversion u64 <-> double happens in the load / store functors.
// S0 > 0: block b of a larger transform whose first S0 stages a strided fp64 pass has done; the hand-off buffer holds
// raw doubles (lazy residues), re-centred on the way in.
template <int LB>
__global__ __launch_bounds__((1 << LB) / 16) void k_ntt_fwd_fp(DevCtx c, u64 *data, const u64 *src, const u64 *src2, size_t split, LimbMap map, int nl, int S0) {
  __shared__ double lds[lds_words(LB)];
  const size_t limb = blockIdx.x >> S0;
  const int b = blockIdx.x & ((1 << S0) - 1);
  const int mid = map.id[limb % nl];
  const Mod m = mod_at(c, mid);
  const FpTable t = fp_table(c, mid);
  u64 *base = data + limb * (size_t)c.n + ((size_t)b << LB);
  const u64 *in = (limb < split ? src + limb * (size_t)c.n : src2 + (limb - split) * (size_t)c.n) + ((size_t)b << LB);
  auto st = [&](int, int i, double v) { base[i] = fp_to_canon(v, m.qd, m.qinv); };
  if (S0 == 0)
    ntt_fwd_block_a<LB, FpArith>(lds, [&](int, int i) { return fp_from_u64(in[i]); }, st, t, m, 0, 0);
  else
    ntt_fwd_block_a<LB, FpArith>(
        lds, [&](int, int i) { return fp_centre(reinterpret_cast<const double *>(in)[i], m.qd, m.qinv); }, st, t, m, S0, b);
}

template <int LB>
__global__ __launch_bounds__((1 << LB) / 16) void k_ntt_inv_fp(DevCtx c, u64 *data, LimbMap map, int nl, int S0) {
  __shared__ double lds[lds_words(LB)];
  const size_t limb = blockIdx.x >> S0;
  const int b = blockIdx.x & ((1 << S0) - 1);
  const int mid = map.id[limb % nl];
  const Mod m = mod_at(c, mid);
  const FpTable t = fp_table(c, mid);
  u64 *base = data + limb * (size_t)c.n + ((size_t)b << LB);
  auto ld = [&](int, int i) { return fp_from_u64(base[i]); };
  if (S0 == 0)
    ntt_inv_block_a<LB, FpArith>(
        lds, ld, [&](int, int i, double v) { base[i] = fp_to_canon(fp_mul_lazy(v, m.inv_n_c, m.inv_n_cq, m.qd), m.qd, m.qinv); }, t, m,
        0, 0);
  else  // the strided pass finishes the transform: hand over raw doubles (|x| <= 16 q after the last block pass)
    ntt_inv_block_a<LB, FpArith>(lds, ld, [&](int, int i, double v) { reinterpret_cast<double *>(base)[i] = v; }, t, m, S0, b);
}

// first R stages of a 2^logn-point forward transform, straight through HBM (coalesced: lane = p)
// (src, src2, split as in the block kernels: limbs [0, split) are read from src, the rest from src2; in place: src == data)
template <int R>
__global__ __launch_bounds__(256) void k_ntt_fwd_strided(DevCtx c, u64 *data, const u64 *src, const u64 *src2, size_t split, LimbMap map,
                                                         int nl) {
  const int G = c.n >> R;
  const int per = G / 256;
  const size_t limb = blockIdx.x / per;
  const int p = (blockIdx.x % per) * 256 + threadIdx.x;
  const int mid = map.id[limb % nl];
  const Mod m = c.mods[mid];
  const NttTable t = ntt_table(c, mid);
  u64 *base = data + limb * (size_t)c.n + p;
  const u64 *in = (limb < split ? src + limb * (size_t)c.n : src2 + (limb - split) * (size_t)c.n) + p;
  u64 x[1 << R];
#pragma unroll
  for (int k = 0; k < (1 << R); k++) x[k] = in[(size_t)k * G];
#pragma unroll
  for (int u = 0; u < R; u++) {
    const int half = 1 << (R - 1 - u);
#pragma unroll
    for (int k = 0; k < (1 << R); k++) {
      if (k & half) continue;
      const int idx = (1 << u) + (k >> (R - u));
      const u64 w = t.tw[idx].x, ws = t.tw[idx].y;
      u64 a = x[k] >= m.two_q ? x[k] - m.two_q : x[k];
      u64 v = mul_shoup_lazy(x[k | half], w, ws, m.q);
      x[k] = a + v;
      x[k | half] = a + m.two_q - v;
    }
  }
#pragma unroll
  for (int k = 0; k < (1 << R); k++) base[(size_t)k * G] = x[k];  // lazy [0,4q): block kernel guards
}

// last R stages of the inverse transform + N^-1 scaling
template <int R>
__global__ __launch_bounds__(256) void k_ntt_inv_strided(DevCtx c, u64 *data, LimbMap map, int nl) {
  const int G = c.n >> R;
  const int per = G / 256;
  const size_t limb = blockIdx.x / per;
  const int p = (blockIdx.x % per) * 256 + threadIdx.x;
  const int mid = map.id[limb % nl];
  const Mod m = c.mods[mid];
  const NttTable t = ntt_table(c, mid);
  u64 *base = data + limb * (size_t)c.n + p;
  u64 x[1 << R];
#pragma unroll
  for (int k = 0; k < (1 << R); k++) x[k] = base[(size_t)k * G];
#pragma unroll
  for (int u = R - 1; u >= 0; u--) {
    const int half = 1 << (R - 1 - u);
#pragma unroll
    for (int k = 0; k < (1 << R); k++) {
      if (k & half) continue;
      const int idx = (1 << u) + (k >> (R - u));
      const u64 w = t.itw[idx].x, ws = t.itw[idx].y;
      u64 a = x[k], b2 = x[k | half];
      u64 s = a + b2;
      x[k] = s >= m.two_q ? s - m.two_q : s;
      x[k | half] = mul_shoup_lazy(a + m.two_q - b2, w, ws, m.q);
    }
  }
#pragma unroll
  for (int k = 0; k < (1 << R); k++) base[(size_t)k * G] = scale_inv_n(x[k], m);
}

// fp64 twins of the two strided passes (every prime of the launch below 2^50).  Forward: u64 in, raw doubles out
// (R <= 4 stages from a canonical input stay below 4.1 q even for 50-bit primes).  Inverse: raw doubles in, re-centred,
// R stages, N^-1, canonical u64 out.
template <int R>
__global__ __launch_bounds__(256) void k_ntt_fwd_strided_fp(DevCtx c, u64 *data, const u64 *src, const u64 *src2, size_t split, LimbMap map,
                                                            int nl) {
  const int G = c.n >> R;
  const int per = G / 256;
  const size_t limb = blockIdx.x / per;
  const int p = (blockIdx.x % per) * 256 + threadIdx.x;
  const int mid = map.id[limb % nl];
  const Mod m = mod_at(c, mid);
  const FpTable t = fp_table(c, mid);
  u64 *base = data + limb * (size_t)c.n + p;
  const u64 *in = (limb < split ? src + limb * (size_t)c.n : src2 + (limb - split) * (size_t)c.n) + p;
  double x[1 << R];
#pragma unroll
  for (int k = 0; k < (1 << R); k++) x[k] = fp_from_u64(in[(size_t)k * G]);
#pragma unroll
  for (int u = 0; u < R; u++) {
    const int half = 1 << (R - 1 - u);
#pragma unroll
    for (int k = 0; k < (1 << R); k++) {
      if (k & half) continue;
      const f64x2 tp = tw_load(t.tw + (1 << u) + (k >> (R - u)));
      const double a = x[k], v = fp_mul_lazy(x[k | half], tp.x, tp.y, m.qd);
      x[k] = a + v;
      x[k | half] = a - v;
    }
  }
#pragma unroll
  for (int k = 0; k < (1 << R); k++) reinterpret_cast<double *>(base)[(size_t)k * G] = x[k];
}
template <int R>
__global__ __launch_bounds__(256) void k_ntt_inv_strided_fp(DevCtx c, u64 *data, LimbMap map, int nl) {
  const int G = c.n >> R;
  const int per = G / 256;
  const size_t limb = blockIdx.x / per;
  const int p = (blockIdx.x % per) * 256 + threadIdx.x;
  const int mid = map.id[limb % nl];
  const Mod m = mod_at(c, mid);
  const FpTable t = fp_table(c, mid);
  u64 *base = data + limb * (size_t)c.n + p;
  double x[1 << R];
#pragma unroll
  for (int k = 0; k < (1 << R); k++) x[k] = fp_centre(reinterpret_cast<const double *>(base)[(size_t)k * G], m.qd, m.qinv);
#pragma unroll
  for (int u = R - 1; u >= 0; u--) {
    const int half = 1 << (R - 1 - u);
#pragma unroll
    for (int k = 0; k < (1 << R); k++) {
      if (k & half) continue;
      const f64x2 tp = tw_load(t.itw + (1 << u) + (k >> (R - u)));
      const double a = x[k], b2 = x[k | half];
      x[k] = a + b2;  // at most 2^R * q/2 = 8 q after R = 4 stages
      x[k | half] = fp_mul_lazy(a - b2, tp.x, tp.y, m.qd);
    }
  }
#pragma unroll
  for (int k = 0; k < (1 << R); k++)
    base[(size_t)k * G] = fp_to_canon(fp_mul_lazy(x[k], m.inv_n_c, m.inv_n_cq, m.qd), m.qd, m.qinv);
}

constexpr int kBigBlockLB = 12;  // LDS block size used under the strided pass for N > 2^14

static bool all_limbs_fp(const abc_hip_ctx *c, const LimbMap &map, int nl) {
  bool fp = c->use_fp;
  for (int j = 0; j < nl; j++) fp = fp && fp_ok(c->h_mods[map.id[j]].bits);
  return fp;
}

template <int LB>
static int launch_block(abc_hip_ctx *c, u64 *d, const LimbMap &map, int nl, size_t total_limbs, int S0, bool fwd,
                        const u64 *src = nullptr, const u64 *src2 = nullptr) {
  if (!src) src = d;
  const size_t split = src2 ? total_limbs / 2 : total_limbs;  // two sources: first half of the limbs from src, second from src2
  dim3 grid((unsigned)(total_limbs << S0)), block((1 << LB) / 16);
  // every limb of the launch must allow the unguarded butterflies.  Behind a strided pre-pass (S0 > 0: guarded, values in
  // [0, 4q)) the at most 12 block stages add 4q each: 52q, inside the 64q the unguarded form is sized for
  bool guard = false;
  const bool fp = all_limbs_fp(c, map, nl);  // with S0 > 0 the strided pass of the same launch makes the same choice
  for (int j = 0; j < nl; j++) guard = guard || !unguarded_ok(c->h_mods[map.id[j]].bits);
  if (fp && fwd)
    hipLaunchKernelGGL(k_ntt_fwd_fp<LB>, grid, block, 0, c->stream, c->dc, d, src, src2, split, map, nl, S0);
  else if (fp)
    hipLaunchKernelGGL(k_ntt_inv_fp<LB>, grid, block, 0, c->stream, c->dc, d, map, nl, S0);
  else if (fwd && guard)
    hipLaunchKernelGGL((k_ntt_fwd<LB, true>), grid, block, 0, c->stream, c->dc, d, src, src2, split, map, nl, S0);
  else if (fwd)
    hipLaunchKernelGGL((k_ntt_fwd<LB, false>), grid, block, 0, c->stream, c->dc, d, src, src2, split, map, nl, S0);
  else
    hipLaunchKernelGGL(k_ntt_inv<LB>, grid, block, 0, c->stream, c->dc, d, map, nl, S0);
  ABC_HIP_CHECK(hipGetLastError());
  return 0;
}

template <int R>
static int launch_strided(abc_hip_ctx *c, u64 *d, const LimbMap &map, int nl, size_t total_limbs, bool fwd, const u64 *src = nullptr,
                          const u64 *src2 = nullptr, bool integer_only = false) {
  const int G = c->n >> R;
  dim3 grid((unsigned)(total_limbs * (G / 256))), block(256);
  const bool fp = !integer_only && all_limbs_fp(c, map, nl);
  if (!src) src = d;
  const size_t split = src2 ? total_limbs / 2 : total_limbs;
  if (fwd && fp)
    hipLaunchKernelGGL(k_ntt_fwd_strided_fp<R>, grid, block, 0, c->stream, c->dc, d, src, src2, split, map, nl);
  else if (fwd)
    hipLaunchKernelGGL(k_ntt_fwd_strided<R>, grid, block, 0, c->stream, c->dc, d, src, src2, split, map, nl);
  else if (fp)
    hipLaunchKernelGGL(k_ntt_inv_strided_fp<R>, grid, block, 0, c->stream, c->dc, d, map, nl);
  else
    hipLaunchKernelGGL(k_ntt_inv_strided<R>, grid, block, 0, c->stream, c->dc, d, map, nl);
  ABC_HIP_CHECK(hipGetLastError());
  return 0;
}

// src != nullptr (forward only): read the coefficients from src, write the transform to d (saves a copy where the
// operand must survive); with a strided pre-pass it is that pass which reads src and writes d.
static int copy_sources(abc_hip_ctx *c, u64 *d, const u64 *src, const u64 *src2, size_t total_limbs) {
  const size_t first = src2 ? total_limbs / 2 : total_limbs;
  ABC_HIP_CHECK(hipMemcpyAsync(d, src, first * (size_t)c->n * 8, hipMemcpyDeviceToDevice, c->stream));
  if (src2)
    ABC_HIP_CHECK(hipMemcpyAsync(d + first * (size_t)c->n, src2, (total_limbs - first) * (size_t)c->n * 8, hipMemcpyDeviceToDevice,
                                 c->stream));
  return 0;
}
static int launch_ntt(abc_hip_ctx *c, u64 *d, const LimbMap &map, int nl, size_t total_limbs, bool fwd, const u64 *src = nullptr,
                      const u64 *src2 = nullptr) {
  if (total_limbs == 0) return 0;
  if (src && !fwd) {
    if (copy_sources(c, d, src, src2, total_limbs)) return 1;
    src = src2 = nullptr;
  }
  switch (c->logn) {
    case 10: return launch_block<10>(c, d, map, nl, total_limbs, 0, fwd, src, src2);
    case 11: return launch_block<11>(c, d, map, nl, total_limbs, 0, fwd, src, src2);
    case 12: return launch_block<12>(c, d, map, nl, total_limbs, 0, fwd, src, src2);
    case 13: return launch_block<13>(c, d, map, nl, total_limbs, 0, fwd, src, src2);
    case 14: {
      // few limbs in flight (single-ciphertext calls): one workgroup per limb leaves most CUs idle for 17-30 us per
      // transform; spread each transform over 64 + 16 workgroups instead (strided radix-16 pass through HBM +
      // 1024-point blocks, the N > 2^14 machinery), trading an HBM round trip nobody misses at this size
      const size_t few = c->sw.few_limbs;
      if (total_limbs > few) return launch_block<14>(c, d, map, nl, total_limbs, 0, fwd, src, src2);
      if (fwd) {
        if (int rc = launch_strided<4>(c, d, map, nl, total_limbs, true, src, src2)) return rc;
        return launch_block<10>(c, d, map, nl, total_limbs, 4, true);
      }
      if (int rc = launch_block<10>(c, d, map, nl, total_limbs, 4, false)) return rc;
      return launch_strided<4>(c, d, map, nl, total_limbs, false);
    }
    case 15:
    case 16: {
      const int S0 = c->logn - kBigBlockLB;
      if (fwd) {
        int rc = (S0 == 3) ? launch_strided<3>(c, d, map, nl, total_limbs, true, src, src2)
                           : launch_strided<4>(c, d, map, nl, total_limbs, true, src, src2);
        if (rc) return rc;
        return launch_block<kBigBlockLB>(c, d, map, nl, total_limbs, S0, true);
      } else {
        int rc = launch_block<kBigBlockLB>(c, d, map, nl, total_limbs, S0, false);
        if (rc) return rc;
        return (S0 == 3) ? launch_strided<3>(c, d, map, nl, total_limbs, false)
                         : launch_strided<4>(c, d, map, nl, total_limbs, false);
      }
    }
    default:
      set_error("unsupported ring degree (logn must be 10..16)");
      return 1;
  }
}

// Key-switch decomposition for N > 2^14, fp64 primes: the strided pass reads operand limb J once per target prime I
// straight from the operand (no reduction modulo q_I is needed in fp64) and writes the half-done limb (ct, J, I), the
// block kernel finishes it -- the separate expand kernel and its round trip disappear.
// dec[ct][J][I] for I <= nl (I = nl: the special prime).  -1: not applicable.
template <int R>
__global__ __launch_bounds__(256) void k_ks_expand_strided_fp(DevCtx c, const u64 *__restrict__ tcoef, size_t tstride,
                                                              u64 *__restrict__ dec, LimbMap map, int nl) {
  const int G = c.n >> R;
  const int per = G / 256;
  const size_t limb = blockIdx.x / per;  // (ct * nl + J) * (nl + 1) + I
  const int p = (blockIdx.x % per) * 256 + threadIdx.x;
  const int I = (int)(limb % (size_t)(nl + 1));
  const size_t cj = limb / (size_t)(nl + 1);
  const int J = (int)(cj % (size_t)nl);
  const size_t ct = cj / (size_t)nl;
  const int mid = map.id[I];
  const Mod m = mod_at(c, mid);
  const FpTable t = fp_table(c, mid);
  const u64 *__restrict__ src = tcoef + ct * tstride + (size_t)J * c.n + p;
  double *__restrict__ dst = reinterpret_cast<double *>(dec) + limb * (size_t)c.n + p;
  double x[1 << R];
#pragma unroll
  for (int k = 0; k < (1 << R); k++) x[k] = fp_from_u64(src[(size_t)k * G]);
#pragma unroll
  for (int u = 0; u < R; u++) {
    const int half = 1 << (R - 1 - u);
#pragma unroll
    for (int k = 0; k < (1 << R); k++) {
      if (k & half) continue;
      const f64x2 tp = tw_load(t.tw + (1 << u) + (k >> (R - u)));
      const double a = x[k], v = fp_mul_lazy(x[k | half], tp.x, tp.y, m.qd);
      x[k] = a + v;
      x[k | half] = a - v;
    }
  }
#pragma unroll
  for (int k = 0; k < (1 << R); k++) dst[(size_t)k * G] = x[k];
}
int launch_ks_expand_ntt_fp(abc_hip_ctx *c, const u64 *tcoef, size_t tstride, u64 *dec, const LimbMap &map, int nl, size_t count) {
  if (c->logn <= 14 || !all_limbs_fp(c, map, nl + 1)) return -1;
  const int S0 = c->logn - kBigBlockLB;
  const size_t limbs = count * nl * (nl + 1);
  const int G = c->n >> S0;
  const dim3 grid((unsigned)(limbs * (G / 256))), block(256);
  if (S0 == 3)
    hipLaunchKernelGGL(k_ks_expand_strided_fp<3>, grid, block, 0, c->stream, c->dc, tcoef, tstride, dec, map, nl);
  else
    hipLaunchKernelGGL(k_ks_expand_strided_fp<4>, grid, block, 0, c->stream, c->dc, tcoef, tstride, dec, map, nl);
  ABC_HIP_CHECK(hipGetLastError());
  return launch_block<kBigBlockLB>(c, dec, map, nl + 1, limbs, S0, true);
}

// N > 2^14: only the strided last stages of the inverse transform (+ N^-1); the block stages were done by a kernel that fused
// them with its own load (abc_kernels_bfv.hip, k_bfv_tensor_inv_block)
// integer_only: the block stages ran in integers whatever the prime (k_iks_special), so the hand-over holds lazy u64 values, not doubles
int launch_ntt_inv_strided_part(abc_hip_ctx *c, u64 *d, const LimbMap &map, int nl, size_t total_limbs, bool integer_only) {
  if (c->logn != 15 && c->logn != 16) { set_error("strided inverse part: N = 2^15 / 2^16 only"); return 1; }
  return (c->logn - kBigBlockLB == 3) ? launch_strided<3>(c, d, map, nl, total_limbs, false, nullptr, nullptr, integer_only)
                                      : launch_strided<4>(c, d, map, nl, total_limbs, false, nullptr, nullptr, integer_only);
}
int big_block_log(void) { return kBigBlockLB; }
// N > 2^14: only the block stages of the forward transform, in place, on limbs whose strided first stages a fused kernel has done
// (abc_kernels_bmul.hip, k_bmul_front): raw doubles in, canonical NTT form out
int launch_ntt_fwd_block_part(abc_hip_ctx *c, u64 *d, const LimbMap &map, int nl, size_t total_limbs) {
  if (c->logn != 15 && c->logn != 16) { set_error("forward block part: N = 2^15 / 2^16 only"); return 1; }
  return launch_block<kBigBlockLB>(c, d, map, nl, total_limbs, c->logn - kBigBlockLB, true);
}

int launch_ntt_fwd(abc_hip_ctx *c, u64 *d, const LimbMap &map, int nl, size_t total_limbs) {
  return launch_ntt(c, d, map, nl, total_limbs, true);
}
int launch_ntt_fwd_from(abc_hip_ctx *c, const u64 *src, u64 *d, const LimbMap &map, int nl, size_t total_limbs) {
  return launch_ntt(c, d, map, nl, total_limbs, true, src);
}
// first half of the limbs from src, second half from src2 (total_limbs even)
int launch_ntt_fwd_from2(abc_hip_ctx *c, const u64 *src, const u64 *src2, u64 *d, const LimbMap &map, int nl, size_t total_limbs) {
  return launch_ntt(c, d, map, nl, total_limbs, true, src, src2);
}
int launch_ntt_inv(abc_hip_ctx *c, u64 *d, const LimbMap &map, int nl, size_t total_limbs) {
  return launch_ntt(c, d, map, nl, total_limbs, false);
}

}  // namespace abc
