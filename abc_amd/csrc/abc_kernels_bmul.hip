// abc_kernels_bmul.hip -- BFV ciphertext x ciphertext multiply (BEHZ) in SPLIT form for N = 2^14: no LDS-resident limb, the
// q -> Bsk extension fused into the forward cross pass, the fast floor + Shenoy-Kumaresan conversion fused into the inverse one,
// and -- for multiply + relinearise -- the key switch's first register pass fused behind the floor.
//
// Reference call sites replaced: src/runtime/SealCiphertext.cpp:104-105,:122-123 (Evaluator::multiply + relinearize_inplace on
// the reference's only scheme and default ring, include/ast_opt/runtime/SealCiphertextFactory.h:13).
//
// A 2^14-point transform = radix-16 cross pass over the 16 values at one position of every 1024-point block + 16 block tails
// (abc_kernels_gsplit.hip).  The BEHZ steps between the transforms are element-wise over COEFFICIENTS and need every limb of a
// coefficient; the cross passes are per LIMB and need 16 coefficients 1024 apart.  A workgroup therefore takes 32 positions x 16
// blocks = 512 coefficients, all limbs, and turns them around in 68 KiB of LDS ([limb][block][position]):
//   M1 k_bmul_front (ct, operand polynomial, position group): read the L q-residues of 512 coefficients, extend to Bsk
//      (fastbconv_m_tilde + sm_mrq, the body of k_behz_extend_fp), park all L + nBsk limbs in LDS, then one thread per
//      (limb, position) runs the forward cross pass -> half-done limbs hA.            32 limbs read, 68 written per pair (L = 8).
//   M2 k_bmul_mid (ct, limb, block; four wavefronts = the four operand polynomials): forward block tails, dyadic tensor product
//      in LDS, inverse block tails of the three product components -> hD.                                      68 read, 51 written.
//   M3 k_bmul_back (ct, component, position group): inverse cross pass + N^-1 of all L + nBsk limbs -> LDS; per coefficient the
//      fast floor and the conversion back to q (the body of k_behz_floor_fp); components 0, 1 -> the result; component 2 stays in
//      LDS and every (digit, position) thread runs the key switch's forward cross pass modulo each key prime -> `part`, what
//      k_fused_operand_pass0_fp<14, false, false> would have produced from a stored c2.                51 read, 16 + 72 written.
// then k_gsplit_special<14, 8, true> / k_bsplit_tcoef / k_bsplit_finish_big (abc_kernels_gsplit.hip) with the result as addend.
// 502 limb transfers per multiply + relinearise where the LDS-resident sequence (bfv_multiply's six kernels + the key switch's
// four) moved ~730 (profiles/r02g_bfv16384_traffic_per_kernel.json: 96 MB per pair).  Same arithmetic as those kernels -- exact
// fp64 residues, every prime (ciphertext and auxiliary) below 2^50 -- bit-identical results (tests/test_gpu_paths.py).
// Other rings, same three kernels: N = 2^13 with four data limbs (BFVDefault(8192): eight blocks, radix-8 cross passes, the whole
// multiply + relinearise sequence); N = 2^15 / 2^16 with eight data limbs (the multiply alone; 32 / 64 blocks of 1024 points, the
// radix-32 / 64 cross passes in two levels through the tile -- a thread holds at most eight values --, rows of 16 / 8 coefficients,
// nine wavefronts; ABC_HIP_NO_BMUL_R6: 4096-point blocks behind radix-8 / 16 register passes, round 3's first form).
#include <algorithm>

#include "abc_context.hpp"

namespace abc {

namespace {

__device__ __forceinline__ double m_mulmod(double x, double y, double q, double qinv) {  // |x|, |y| <= q -> |result| < q
  const double h = x * y;
  const double l = __builtin_fma(x, y, -h);
  return __builtin_fma(-__builtin_rint(h * qinv), q, h) + l;
}
__device__ __forceinline__ double m_canon_d(double x, double q, double qinv) {  // any lazy value -> canonical [0, q) as a double
  const double r = fp_centre(x, q, qinv);
  return r < 0.0 ? r + q : r;
}
// constant rows fetched inside their own iteration (see abc_kernels_bfv.hip, row_after)
template <class T>
__device__ __forceinline__ const ABC_CONST_AS T *m_row_after(const ABC_CONST_AS T *p, u64 dep) {
  asm volatile("" : "+s"(p) : "v"(dep));
  return p;
}
// per-LANE modulus constants (a wavefront of the cross-pass phases works on two limbs)
struct LaneMod {
  FpK kk;
  double inv_n_c, inv_n_cq;
  FpTable t;
};
__device__ __forceinline__ LaneMod lane_mod(const DevCtx &c, int mid) {
  const Mod *p = c.mods + mid;
  LaneMod r;
  r.kk.q = p->qd;
  r.kk.qinv = p->qinv;
  r.kk.red = p->bits >= 49;
  r.inv_n_c = p->inv_n_c;
  r.inv_n_cq = p->inv_n_cq;
  r.t = fp_table(c, mid);
  return r;
}

// radix-2^R cross passes on the 2^R values of one column; R = 4 through the register pass of abc_ntt.hpp (the form the N = 2^14
// key-switch kernels use)
template <int R>
__device__ __forceinline__ void m_fwd_cross(double (&x)[1 << R], const FpTable &t, const FpK &kk) {
  if constexpr (R == 4) {
    const int hi0[1] = {0};
    fwd_pass<FpArith, 14, 0, 4>(x, hi0, t, kk, 0, 0);
  } else {
    fwd_cross<R>(x, t, kk);
  }
}
template <int R>
__device__ __forceinline__ void m_inv_cross(double (&x)[1 << R], const FpTable &t, const FpK &kk) {
  if constexpr (R == 4) {
    const int hi0[1] = {0};
    inv_pass<FpArith, 14, 0, 4>(x, hi0, t, kk, 0, 0);
  } else {
    inv_cross<R>(x, t, kk);
  }
}

// tile [limb][block][position] of M1 / M3.  R > 4 (two-level cross passes): one position group of padding after every eight
// block rows -- the second level's lanes are (group of eight blocks, position), and unpadded their rows lie 8 P doubles apart: on
// the same LDS banks.
template <int R, int NBLK, int P>
__device__ __forceinline__ int m_tix(int l, int blk, int p) {
  const int row = l * NBLK + blk;
  return (R > 4 ? row + (row >> 3) : row) * P + p;
}

}  // namespace

// ---- M1 ----
// LOGN, R: ring and radix of the cross pass: the 2^R values one cross pass takes are N >> R apart (N = 2^14: R = 4 over 1024-point
// blocks; N = 2^15 / 2^16: R = 3 / 4 over the 4096-point blocks of the big-ring transforms, abc_kernels_ntt.hip).  A workgroup
// takes 512 coefficients: 2^R blocks x P = 512 >> R positions.
template <int LOGN, int R, int LT, int NBT, int NT = 512>
__global__ __launch_bounds__(NT, 4) void k_bmul_front(DevCtx c, const u64 *__restrict__ a, const u64 *__restrict__ b, double *__restrict__ hA) {
  constexpr int L = LT, nBsk = NBT + 1, NLM = L + nBsk;
  constexpr int NBLK = 1 << R, P = 512 >> R, LOGP = 9 - R, SH = LOGN - R, NPG = (1 << SH) / P;
  constexpr size_t N = (size_t)1 << LOGN;
  extern __shared__ double dyn[];  // [NLM][NBLK][P]
  const ABC_CONST_AS DevConst &k = *(const ABC_CONST_AS DevConst *)c.cst;
  const ABC_CONST_AS DevConstFp &f = *(const ABC_CONST_AS DevConstFp *)c.cstf;
  // R > 4: a workgroup's rows in HBM are P = 8 or 16 coefficients (half a 128-byte line or one): neighbouring position groups go to
  // the same XCD, back to back, so the other half of a line is found in that XCD's L2 (consecutive ids alternate over the eight XCDs)
  const unsigned wid = R > 4 ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  const int pg = (int)(wid % (unsigned)NPG);
  const int poly = (int)((wid / (unsigned)NPG) & 3u);  // a0, a1, b0, b1
  const size_t ct = wid / (unsigned)(NPG * 4);
  const int tid = threadIdx.x;
  const size_t pw = (size_t)L * N;
  const u64 *__restrict__ src = (poly < 2 ? a : b) + ct * 2 * pw + (size_t)(poly & 1) * pw;
  if (NT == 512 || tid < 512) {  // one coefficient per thread: (block kb, position p); NT = 576: a ninth wavefront for the cross passes
    const int kb = tid >> LOGP, p = tid & (P - 1);
    const size_t x = ((size_t)kb << SH) + (size_t)(pg * P) + p;
    u64 raw[L];
#pragma unroll
    for (int i = 0; i < L; i++) raw[i] = src[(size_t)i * N + x];
    double tmp[L];
    u32 mt = 0;
#pragma unroll
    for (int i = 0; i < L; i++) {
      const Mod m = mod_at(c, i);
      const double v0 = fp_from_u64(raw[i]);
      dyn[m_tix<R, NBLK, P>(i, kb, p)] = v0;
      const double v = m_canon_d(fp_mul_lazy(v0, f.ext_q[i][0], f.ext_q[i][1], m.qd), m.qd, m.qinv);
      tmp[i] = v;
      mt += (u32)(u64)__double_as_longlong(v + 4503599627370496.0) * (u32)k.q_to_mtilde[i];  // mod 2^32 on the canonical residue
    }
    const u32 r32 = mt * (u32)k.neg_inv_q_mod_mtilde;
    const double r = (double)(int)r32;  // centred representative of r mod m~
    u64 dep = (u64)r32;
#pragma unroll
    for (int j = 0; j < nBsk; j++) {
      const Mod m = mod_at(c, c.id_bsk + j);
      const ABC_CONST_AS double *row = m_row_after(&f.q_to_bsk[j][0][0], dep);
      double conv = fp_mul_lazy(r, f.q_mod_bsk[j][0], f.q_mod_bsk[j][1], m.qd);
#pragma unroll
      for (int i = 0; i < L; i++) conv += fp_mul_lazy(tmp[i], row[2 * i], row[2 * i + 1], m.qd);
      const double v = m_canon_d(fp_mul_lazy(conv, f.inv_mtilde_mod_bsk[j][0], f.inv_mtilde_mod_bsk[j][1], m.qd), m.qd, m.qinv);
      dep = (u64)__double_as_longlong(v);
      dyn[m_tix<R, NBLK, P>((L + j), kb, p)] = v;
    }
  }
  __syncthreads();
  if constexpr (R > 4) {
    // radix-2^R cross pass in two levels through the tile (2^R values per column do not fit a thread: 32 values ran the inverse
    // pass at half speed): stages 0..RB-1 on the 2^RB blocks 8 apart, in place; stages RB..R-1 on 8
    // consecutive blocks, out to hA.  Jobs are (limb, group, position) with the position fastest: 8 P lanes = one limb per wavefront.
    constexpr int RB = R - 3, NH = 1 << RB;
    for (int job = tid; job < NLM * 8 * P; job += NT) {
      const int p = job & (P - 1), g = (job >> LOGP) & 7;
      const int l = __builtin_amdgcn_readfirstlane(job >> (LOGP + 3));  // 8 P = 64 or 128 jobs per limb: wavefront-uniform
      const LaneMod lm = lane_mod(c, l < L ? l : c.id_bsk + (l - L));
      double y[NH];
#pragma unroll
      for (int h = 0; h < NH; h++) y[h] = dyn[m_tix<R, NBLK, P>(l, (h << 3) + g, p)];
#pragma unroll
      for (int u = 0; u < RB; u++) {
        const int hf = 1 << (RB - 1 - u);
#pragma unroll
        for (int h = 0; h < NH; h++) {
          if (h & hf) continue;
          FpArith::fwd(y[h], y[h | hf], tw_load(lm.t.tw + (1 << u) + (h >> (RB - u))), lm.kk);
        }
      }
#pragma unroll
      for (int h = 0; h < NH; h++) dyn[m_tix<R, NBLK, P>(l, (h << 3) + g, p)] = y[h];
    }
    __syncthreads();
    for (int job = tid; job < NLM * NH * P; job += NT) {
      const int p = job & (P - 1), jg = (job >> LOGP) & (NH - 1);
      const int l = __builtin_amdgcn_readfirstlane(job >> (LOGP + RB));
      const LaneMod lm = lane_mod(c, l < L ? l : c.id_bsk + (l - L));
      double v[8];
#pragma unroll
      for (int j = 0; j < 8; j++) v[j] = dyn[m_tix<R, NBLK, P>(l, (jg << 3) + j, p)];
#pragma unroll
      for (int u = RB; u < R; u++) {
        const int hf = 1 << (R - 1 - u);
#pragma unroll
        for (int j = 0; j < 8; j++) {
          if (j & hf) continue;
          FpArith::fwd(v[j], v[j | hf], tw_load(lm.t.tw + (1 << u) + (((jg << 3) + j) >> (R - u))), lm.kk);
        }
      }
      double *__restrict__ dst = hA + ((ct * 4 + poly) * NLM + l) * N + (size_t)(pg * P) + p;
#pragma unroll
      for (int j = 0; j < 8; j++) dst[(size_t)((jg << 3) + j) << SH] = v[j];
    }
  } else {
    for (int job = tid; job < NLM * P; job += 512) {  // one (limb, position) column per thread; the last limb is a second round
      const int l = job >> LOGP, p = job & (P - 1);
      const LaneMod lm = lane_mod(c, l < L ? l : c.id_bsk + (l - L));
      double x[NBLK];
#pragma unroll
      for (int kb = 0; kb < NBLK; kb++) x[kb] = dyn[m_tix<R, NBLK, P>(l, kb, p)];
      m_fwd_cross<R>(x, lm.t, lm.kk);
      double *__restrict__ dst = hA + ((ct * 4 + poly) * NLM + l) * N + (size_t)(pg * P) + p;
#pragma unroll
      for (int kb = 0; kb < NBLK; kb++) dst[(size_t)kb << SH] = x[kb];
    }
  }
}

// ---- M2 ----
// LB: block size of the tails (10 at N = 2^14: one wavefront per operand polynomial, 256 threads, 35 KiB; 12 at N = 2^15 / 2^16:
// 256 threads per operand polynomial, 1024 threads, 139 KiB -- one workgroup per CU, but its 187 limb transfers per limb-block
// shrink to 7: the separate forward-tail kernel wrote and the tensor kernel re-read every operand limb).  Groups of T = 2^LB / 16
// threads run the block transforms of abc_ntt.hpp side by side; for LB > 10 those contain one workgroup barrier each, which the
// fourth group -- it has no product component to transform back -- joins without doing the work.
template <int LOGN, int LB>
__global__ __launch_bounds__(4 * ((1 << LB) / 16)) void k_bmul_mid(DevCtx c, const double *__restrict__ hA, double *__restrict__ hD, int nlm, int L) {
  constexpr int LOGNB = LOGN - LB, T = (1 << LB) / 16, LW = lds_words(LB);
  constexpr size_t N = (size_t)1 << LOGN;
  extern __shared__ double dyn[];  // four transform buffers
  const int G = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / T));  // operand polynomial / product component of this group
  const int gt = threadIdx.x & (T - 1);
  const int blk = blockIdx.x & ((1 << LOGNB) - 1);
  const int l = (int)((blockIdx.x >> LOGNB) % (unsigned)nlm);
  const size_t ct = (size_t)((blockIdx.x >> LOGNB) / (unsigned)nlm);
  const int mid = l < L ? l : c.id_bsk + (l - L);
  const Mod m = mod_at(c, mid);
  const FpTable t = fp_table(c, mid);
  const double q = m.qd, qinv = m.qinv;
  const size_t base = (size_t)blk << LB;
  double *buf = dyn + G * LW;
  {  // group G: forward tail of operand polynomial G (a0, a1, b0, b1), centred result parked in LDS
    const double *__restrict__ src = hA + ((ct * 4 + G) * nlm + l) * N + base;
    ntt_fwd_block_a<LB, FpTail>(
        buf, [&](int, int i) { return fp_centre(src[i], q, qinv); }, [&](int, int i, double v) { buf[lds_pad(i)] = fp_centre(v, q, qinv); }, t,
        m, LOGNB, blk, gt);
  }
  __syncthreads();
  for (int e = 2 * (int)threadIdx.x; e < (1 << LB); e += 8 * T) {  // dyadic tensor product, in place (a thread rewrites only what it read)
    double *p0 = dyn + lds_pad(e), *p1 = p0 + LW, *p2 = p1 + LW, *p3 = p2 + LW;
    const f64x2 a0 = *reinterpret_cast<const f64x2 *>(p0), a1 = *reinterpret_cast<const f64x2 *>(p1);
    const f64x2 b0 = *reinterpret_cast<const f64x2 *>(p2), b1 = *reinterpret_cast<const f64x2 *>(p3);
    f64x2 d;
    d.x = m_mulmod(a0.x, b0.x, q, qinv);
    d.y = m_mulmod(a0.y, b0.y, q, qinv);
    *reinterpret_cast<f64x2 *>(p0) = d;
    d.x = fp_centre(m_mulmod(a0.x, b1.x, q, qinv) + m_mulmod(a1.x, b0.x, q, qinv), q, qinv);
    d.y = fp_centre(m_mulmod(a0.y, b1.y, q, qinv) + m_mulmod(a1.y, b0.y, q, qinv), q, qinv);
    *reinterpret_cast<f64x2 *>(p1) = d;
    d.x = m_mulmod(a1.x, b1.x, q, qinv);
    d.y = m_mulmod(a1.y, b1.y, q, qinv);
    *reinterpret_cast<f64x2 *>(p2) = d;
  }
  __syncthreads();
  if (G < 3) {  // inverse tail of product component G; N^-1 belongs to the cross pass (M3)
    double *tb = dyn + G * LW;
    double *__restrict__ dst = hD + ((ct * 3 + G) * nlm + l) * N + base;
    ntt_inv_block_a<LB, FpArith>(
        tb, [&](int, int i) { return tb[lds_pad(i)]; }, [&](int, int i, double v) { dst[i] = v; }, t, m, LOGNB, blk, gt);
  } else if (LB > 10) {
    block_sync_lds();  // the ONE workgroup barrier inside ntt_inv_block_a<LB > 10> (before its last pass): the fourth group only keeps count
  }
}

// ---- M3 ----
// want3 = 1: plain multiply, all three components to out [ct][3][L][N], no key-switch pass (the only form for N > 2^14, whose key
// switch decomposes over 1024-point blocks: abc_kernels_gsplit.hip, bsplit_big)
template <int LOGN, int R, int LT, int NBT, int NT = 512>
__global__ __launch_bounds__(NT, 4) void k_bmul_back(DevCtx c, const double *__restrict__ hD, u64 *__restrict__ out, double *__restrict__ part,
                                                      int want3) {
  constexpr int L = LT, nB = NBT, nBsk = NBT + 1, NLM = L + nBsk;
  constexpr int NBLK = 1 << R, P = 512 >> R, LOGP = 9 - R, SH = LOGN - R, NPG = (1 << SH) / P;
  constexpr size_t N = (size_t)1 << LOGN;
  extern __shared__ double dyn[];  // [NLM][NBLK][P]
  const ABC_CONST_AS DevConstFp &f = *(const ABC_CONST_AS DevConstFp *)c.cstf;
  const unsigned wid = R > 4 ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;  // as in k_bmul_front
  const int pg = (int)(wid % (unsigned)NPG);
  const int comp = (int)((wid / (unsigned)NPG) % 3u);
  const size_t ct = (size_t)((wid / (unsigned)NPG) / 3u);
  const int tid = threadIdx.x;
  if constexpr (R > 4) {  // two levels through the tile (see k_bmul_front): stages R-1..RB on 8 consecutive blocks, then RB-1..0 on blocks 8 apart
    constexpr int RB = R - 3, NH = 1 << RB;
    for (int job = tid; job < NLM * NH * P; job += NT) {
      const int p = job & (P - 1), jg = (job >> LOGP) & (NH - 1);
      const int l = __builtin_amdgcn_readfirstlane(job >> (LOGP + RB));
      const LaneMod lm = lane_mod(c, l < L ? l : c.id_bsk + (l - L));
      const double *__restrict__ src = hD + ((ct * 3 + comp) * NLM + l) * N + (size_t)(pg * P) + p;
      double v[8];
#pragma unroll
      for (int j = 0; j < 8; j++) v[j] = fp_centre(src[(size_t)((jg << 3) + j) << SH], lm.kk.q, lm.kk.qinv);
#pragma unroll
      for (int u = R - 1; u >= RB; u--) {
        const int hf = 1 << (R - 1 - u);
#pragma unroll
        for (int j = 0; j < 8; j++) {
          if (j & hf) continue;
          FpArith::inv(v[j], v[j | hf], tw_load(lm.t.itw + (1 << u) + (((jg << 3) + j) >> (R - u))), lm.kk);
        }
      }
#pragma unroll
      for (int j = 0; j < 8; j++) dyn[m_tix<R, NBLK, P>(l, (jg << 3) + j, p)] = fp_centre(v[j], lm.kk.q, lm.kk.qinv);
    }
    __syncthreads();
    for (int job = tid; job < NLM * 8 * P; job += NT) {
      const int p = job & (P - 1), g = (job >> LOGP) & 7;
      const int l = __builtin_amdgcn_readfirstlane(job >> (LOGP + 3));  // 8 P = 64 or 128 jobs per limb: wavefront-uniform
      const LaneMod lm = lane_mod(c, l < L ? l : c.id_bsk + (l - L));
      double x[NH];
#pragma unroll
      for (int h = 0; h < NH; h++) x[h] = dyn[m_tix<R, NBLK, P>(l, (h << 3) + g, p)];
#pragma unroll
      for (int u = RB - 1; u >= 0; u--) {
        const int hf = 1 << (RB - 1 - u);
#pragma unroll
        for (int h = 0; h < NH; h++) {
          if (h & hf) continue;
          FpArith::inv(x[h], x[h | hf], tw_load(lm.t.itw + (1 << u) + (h >> (RB - u))), lm.kk);
        }
      }
#pragma unroll
      for (int h = 0; h < NH; h++) dyn[m_tix<R, NBLK, P>(l, (h << 3) + g, p)] = fp_mul_lazy(x[h], lm.inv_n_c, lm.inv_n_cq, lm.kk.q);
    }
  } else {
    for (int job = tid; job < NLM * P; job += 512) {  // inverse cross pass + N^-1 of one (limb, position) column
      const int l = job >> LOGP, p = job & (P - 1);
      const LaneMod lm = lane_mod(c, l < L ? l : c.id_bsk + (l - L));
      const double *__restrict__ src = hD + ((ct * 3 + comp) * NLM + l) * N + (size_t)(pg * P) + p;
      double x[NBLK];
#pragma unroll
      for (int kb = 0; kb < NBLK; kb++) x[kb] = fp_centre(src[(size_t)kb << SH], lm.kk.q, lm.kk.qinv);
      m_inv_cross<R>(x, lm.t, lm.kk);
#pragma unroll
      for (int kb = 0; kb < NBLK; kb++) dyn[m_tix<R, NBLK, P>(l, kb, p)] = fp_mul_lazy(x[kb], lm.inv_n_c, lm.inv_n_cq, lm.kk.q);
    }
  }
  __syncthreads();
  static_assert(NT == 512 || LOGN > 14, "the ninth wavefront leaves here: only where no later phase needs a barrier");
  if (NT > 512 && tid >= 512) return;
  {  // BEHZ steps (6)-(8) on one coefficient per thread (k_behz_floor_fp): scale by t, fast floor by q, Shenoy-Kumaresan back to q
    const int kb = tid >> LOGP, p = tid & (P - 1);
    const size_t x = ((size_t)kb << SH) + (size_t)(pg * P) + p;
    const Mod msk = mod_at(c, c.id_bsk + nB);
    double tq[L], fl[nBsk];
#pragma unroll
    for (int i = 0; i < L; i++) {
      const Mod m = mod_at(c, i);
      tq[i] = m_canon_d(fp_mul_lazy(dyn[m_tix<R, NBLK, P>(i, kb, p)], f.flr_q[i][0], f.flr_q[i][1], m.qd), m.qd, m.qinv);
    }
    u64 dep = (u64)__double_as_longlong(tq[L - 1]);
#pragma unroll
    for (int j = 0; j < nBsk; j++) {
      const Mod m = mod_at(c, c.id_bsk + j);
      const ABC_CONST_AS double *row = m_row_after(&f.q_to_bsk[j][0][0], dep);
      double conv = 0.0;
#pragma unroll
      for (int i = 0; i < L; i++) conv += fp_mul_lazy(tq[i], row[2 * i], row[2 * i + 1], m.qd);
      const double xb = fp_mul_lazy(dyn[m_tix<R, NBLK, P>((L + j), kb, p)], f.tinvq_bsk[j][0], f.tinvq_bsk[j][1], m.qd);
      fl[j] = xb - fp_mul_lazy(conv, f.inv_q_mod_bsk[j][0], f.inv_q_mod_bsk[j][1], m.qd);
      dep = (u64)__double_as_longlong(fl[j]);
    }
    double tb[nB];
#pragma unroll
    for (int b2 = 0; b2 < nB; b2++) {
      const Mod m = mod_at(c, c.id_bsk + b2);
      tb[b2] = m_canon_d(fp_mul_lazy(fl[b2], f.inv_punct_B[b2][0], f.inv_punct_B[b2][1], m.qd), m.qd, m.qinv);
    }
    double mconv = -fl[nB];
#pragma unroll
    for (int b2 = 0; b2 < nB; b2++) mconv += fp_mul_lazy(tb[b2], f.B_to_msk[b2][0], f.B_to_msk[b2][1], msk.qd);
    double alpha = m_canon_d(fp_mul_lazy(mconv, f.inv_B_mod_msk[0], f.inv_B_mod_msk[1], msk.qd), msk.qd, msk.qinv);
    if (alpha > (double)(msk.q >> 1)) alpha -= msk.qd;
    dep = (u64)__double_as_longlong(alpha);
    const bool to_out = want3 || comp < 2;  // workgroup-uniform
    u64 *__restrict__ o = out + (ct * (want3 ? 3 : 2) + comp) * (size_t)L * N + x;
#pragma unroll
    for (int i = 0; i < L; i++) {
      const Mod m = mod_at(c, i);
      const ABC_CONST_AS double *row = m_row_after(&f.B_to_q[i][0][0], dep);
      double v = -fp_mul_lazy(alpha, f.B_mod_q[i][0], f.B_mod_q[i][1], m.qd);
#pragma unroll
      for (int b2 = 0; b2 < nB; b2++) v += fp_mul_lazy(tb[b2], row[2 * b2], row[2 * b2 + 1], m.qd);
      if (to_out) {
        const u64 w = fp_to_canon(v, m.qd, m.qinv);
        dep = w;
        o[(size_t)i * N] = w;
      } else {  // c2: canonical [0, q_i) as a double, the value the key switch's decomposition reduces modulo the other primes
        const double w = m_canon_d(v, m.qd, m.qinv);
        dep = (u64)__double_as_longlong(w);
        dyn[m_tix<R, NBLK, P>(i, kb, p)] = w;
      }
    }
    if (to_out) return;
  }
  if constexpr (LOGN <= 14) {
    __syncthreads();
    // key switch, first step: digit J of c2 at this position group, forward cross pass modulo every key prime I (two halves of
    // the key primes on two sets of wavefronts) -> part[ct][I][J], the layout k_gsplit_special<14, L, true> reads
    const size_t PS = (size_t)c.ps;
    for (int job = tid; job < L * P * 2; job += 512) {
      const int p = job & (P - 1), J = (job >> LOGP) % L;
      const int half = __builtin_amdgcn_readfirstlane((job >> LOGP) / L);  // wave-uniform: a wavefront covers two J of one half (L even)
      const int I0 = half ? (L + 2) / 2 : 0, I1 = half ? L + 1 : (L + 2) / 2;
      double x[NBLK];
#pragma unroll
      for (int kb = 0; kb < NBLK; kb++) x[kb] = dyn[m_tix<R, NBLK, P>(J, kb, p)];
      for (int I = I0; I < I1; I++) {
        const int ki = (I == L) ? c.K - 1 : I;
        const Mod m = mod_at(c, ki);
        const FpTable t = fp_table(c, ki);
        const FpK kk = FpArith::consts(m);
        double y[NBLK];
#pragma unroll
        for (int kb = 0; kb < NBLK; kb++) y[kb] = x[kb];
        m_fwd_cross<R>(y, t, kk);
        double *__restrict__ dst = part + ((ct * (L + 1) + I) * L + J) * PS + (size_t)(pg * P) + p;
#pragma unroll
        for (int kb = 0; kb < NBLK; kb++) dst[(size_t)kb << SH] = y[kb];
      }
    }
  }
}

// ---- host side ----
static bool bmul_shape(const abc_hip_ctx *c, int limbs = 8) {
  return c->scheme == 1 && c->use_fp && c->behz_fp && !c->sw.no_bmul && !c->sw.no_split && !c->sw.no_fused && c->L == limbs &&
         c->nB == limbs && c->K == c->L + 1;
}
// multiply + relinearise in one sequence (N = 2^14)
bool bmul_applies(const abc_hip_ctx *c) {
  if (c->logn == 13) return bmul_shape(c, 4) && bsplit_big_applies(c, c->L);  // BFVDefault(8192)
  return c->logn == 14 && bmul_shape(c) && bsplit_applies(c, c->L);
}
// the multiply alone (size-3 product): also N = 2^15 / 2^16 over the 4096-point blocks of the big-ring transforms
bool bmul_multiply_applies(const abc_hip_ctx *c) {
  if (c->logn == 14) return bmul_applies(c);
  if (c->logn == 13) return bmul_shape(c, 4);  // BFVDefault(8192): four data limbs, radix-8 cross passes over eight 1024-point blocks
  return (c->logn == 15 || c->logn == 16) && bmul_shape(c) && !c->sw.no_gsplit && big_block_log() == 12;
}

// scratch per ciphertext pair (words): X = max(hA, part) | Y = max(hD, half + tco)
static size_t bmul_scratch_words(const abc_hip_ctx *c) {
  const size_t N = (size_t)c->n, PS = (size_t)c->dc.ps;
  const int L = c->L, nlm = c->L + c->nBsk;
  const size_t X = std::max((size_t)4 * nlm * N, (size_t)L * (L + 1) * PS);
  const size_t Y = std::max((size_t)3 * nlm * N, (size_t)(2 * (L + 1) + 2) * PS);
  return X + Y;
}

// N = 2^15 / 2^16: M1, the block tails of the forward transforms (k_ntt_fwd_fp<12>), the tensor product inside the block tails of
// the inverse ones (k_bfv_tensor_inv_block<12>), M3 -- where the generic sequence ran extend, two strided + two block forward
// passes, two tensor / block-inverse launches, two strided inverse passes and the floor (704 limb transfers per pair; here 498)
static int bmul_big(abc_hip_ctx *c, const u64 *a, const u64 *b, u64 *out3, size_t count) {
  const size_t N = (size_t)c->n;
  const int L = c->L, nlm = c->L + c->nBsk;
  const size_t per_ct = (size_t)7 * nlm * N;
  size_t chunk = (((size_t)2 << 30) / 8) / per_ct;  // 4 and 8 GiB measured: no difference (config 5: 1 460 circuits/s each)
  if (c->sw.bfv_scratch_mb) chunk = (c->sw.bfv_scratch_mb << 20) / 8 / per_ct;
  if (chunk < 1) chunk = 1;
  if (chunk > count) chunk = count;
  else if (count % chunk && count / chunk < 8) chunk = (count + count / chunk) / (count / chunk + 1);  // even chunks, no runt
  if (ensure_workspace(c, chunk * per_ct * 8)) return 1;
  LimbMap map{};
  for (int j = 0; j < L; j++) map.id[j] = j;
  for (int j = 0; j < c->nBsk; j++) map.id[L + j] = c->dc.id_bsk + j;
  const size_t lds = (size_t)nlm * 512 * 8;
  hipStream_t st = c->stream;
  for (size_t off = 0; off < count; off += chunk) {
    const size_t cc = (count - off < chunk) ? count - off : chunk;
    double *X = (double *)c->ws, *Y = X + cc * 4 * nlm * N;
    const u64 *pa = a + off * 2 * L * N, *pb = b + off * 2 * L * N;
    u64 *po = out3 + off * 3 * L * N;
    if (c->logn == 13) {
      hipLaunchKernelGGL((k_bmul_front<13, 3, 4, 4>), dim3((unsigned)(cc * 4 * 16)), dim3(512), lds, st, c->dc, pa, pb, X);
      hipLaunchKernelGGL((k_bmul_mid<13, 10>), dim3((unsigned)(cc * nlm * 8)), dim3(256), (size_t)(4 * lds_words(10)) * 8, st, c->dc,
                         (const double *)X, Y, nlm, L);
      hipLaunchKernelGGL((k_bmul_back<13, 3, 4, 4>), dim3((unsigned)(cc * 3 * 16)), dim3(512), lds, st, c->dc, (const double *)Y, po, X, 1);
      ABC_HIP_CHECK(hipGetLastError());
      continue;
    }
    if (c->logn == 16 && !c->sw.no_bmul_r6 && !c->sw.no_bmul_mid) {  // 1024-point blocks: M2 as four workgroups per CU, radix-64 cross passes in two levels
      const size_t lds = (size_t)nlm * 576 * 8;  // padded tile (m_tix)
      hipLaunchKernelGGL((k_bmul_front<16, 6, 8, 8, 576>), dim3((unsigned)(cc * 4 * 128)), dim3(576), lds, st, c->dc, pa, pb, X);
      hipLaunchKernelGGL((k_bmul_mid<16, 10>), dim3((unsigned)(cc * nlm * 64)), dim3(256), (size_t)(4 * lds_words(10)) * 8, st, c->dc,
                         (const double *)X, Y, nlm, L);
      hipLaunchKernelGGL((k_bmul_back<16, 6, 8, 8, 576>), dim3((unsigned)(cc * 3 * 128)), dim3(576), lds, st, c->dc, (const double *)Y, po, X, 1);
      ABC_HIP_CHECK(hipGetLastError());
      continue;
    }
    if (c->logn == 15 && !c->sw.no_bmul_r6 && !c->sw.no_bmul_mid) {  // the same at N = 2^15: radix-32 cross passes as 4 x 8, rows of 16 coefficients
      const size_t lds = (size_t)nlm * 576 * 8;
      hipLaunchKernelGGL((k_bmul_front<15, 5, 8, 8, 576>), dim3((unsigned)(cc * 4 * 64)), dim3(576), lds, st, c->dc, pa, pb, X);
      hipLaunchKernelGGL((k_bmul_mid<15, 10>), dim3((unsigned)(cc * nlm * 32)), dim3(256), (size_t)(4 * lds_words(10)) * 8, st, c->dc,
                         (const double *)X, Y, nlm, L);
      hipLaunchKernelGGL((k_bmul_back<15, 5, 8, 8, 576>), dim3((unsigned)(cc * 3 * 64)), dim3(576), lds, st, c->dc, (const double *)Y, po, X, 1);
      ABC_HIP_CHECK(hipGetLastError());
      continue;
    }
    if (c->logn == 16)
      hipLaunchKernelGGL((k_bmul_front<16, 4, 8, 8>), dim3((unsigned)(cc * 4 * 128)), dim3(512), lds, st, c->dc, pa, pb, X);
    else
      hipLaunchKernelGGL((k_bmul_front<15, 3, 8, 8>), dim3((unsigned)(cc * 4 * 64)), dim3(512), lds, st, c->dc, pa, pb, X);
    ABC_HIP_CHECK(hipGetLastError());
    if (c->sw.no_bmul_mid) {  // A/B: block tails and tensor product as the separate kernels of the generic sequence
      if (launch_ntt_fwd_block_part(c, (u64 *)X, map, nlm, cc * 4 * nlm)) return 1;
      if (launch_bfv_tensor_inv_block(c, (const u64 *)X, (const u64 *)X + 2 * (size_t)nlm * N, 4 * (size_t)nlm * N, (u64 *)Y, map, nlm, cc))
        return 1;
    } else if (c->logn == 16) {
      hipLaunchKernelGGL((k_bmul_mid<16, 12>), dim3((unsigned)(cc * nlm * 16)), dim3(1024), (size_t)(4 * lds_words(12)) * 8, st, c->dc,
                         (const double *)X, Y, nlm, L);
    } else {
      hipLaunchKernelGGL((k_bmul_mid<15, 12>), dim3((unsigned)(cc * nlm * 8)), dim3(1024), (size_t)(4 * lds_words(12)) * 8, st, c->dc,
                         (const double *)X, Y, nlm, L);
    }
    if (c->logn == 16)
      hipLaunchKernelGGL((k_bmul_back<16, 4, 8, 8>), dim3((unsigned)(cc * 3 * 128)), dim3(512), lds, st, c->dc, (const double *)Y, po, X, 1);
    else
      hipLaunchKernelGGL((k_bmul_back<15, 3, 8, 8>), dim3((unsigned)(cc * 3 * 64)), dim3(512), lds, st, c->dc, (const double *)Y, po, X, 1);
    ABC_HIP_CHECK(hipGetLastError());
  }
  return 0;
}

// relin = true: out [count][2][L][N] = relinearised product; false: out [count][3][L][N] = the size-3 product
int bmul_split(abc_hip_ctx *c, const u64 *a, const u64 *b, u64 *out, size_t count, bool relin) {
  if (!count) return 0;
  if (c->logn > 14 || (c->logn == 13 && !relin)) {
    if (relin) { set_error("bmul_split: multiply + relinearise in one sequence is an N = 2^13 / 2^14 path"); return 1; }
    return bmul_big(c, a, b, out, count);
  }
  const bool n13 = c->logn == 13;
  const size_t N = (size_t)c->n, PS = (size_t)c->dc.ps;
  const int L = c->L, nlm = c->L + c->nBsk;
  const size_t per_ct = bmul_scratch_words(c);
  // chunks alternate over the context's internal lanes, as in the CKKS hot call (plan_chunks, abc_kernels_fused.hip)
  int lanes = c->sw.lanes;
  if (count <= 8 || lanes < 1) lanes = 1;
  size_t chunk = c->sw.chunk ? c->sw.chunk : 64;
  if (c->sw.bfv_scratch_mb) chunk = std::max<size_t>(1, (c->sw.bfv_scratch_mb << 20) / 8 / per_ct / (size_t)lanes);
  if (chunk * lanes > count) chunk = (count + lanes - 1) / lanes;
  if (ensure_workspace(c, (size_t)lanes * chunk * per_ct * 8)) return 1;
  const size_t Xw = std::max((size_t)4 * nlm * N, (size_t)L * (L + 1) * PS);
  const size_t lds = (size_t)nlm * 512 * 8;
  if (relin) (void)key_twin(c, c->d_relin);  // before the lanes fork (the inner-product kernel reads it)
  LaneScope scope(c, lanes);
  if (scope.fork()) return 1;
  int turn = 0;
  for (size_t off = 0; off < count; off += chunk, turn++) {
    const size_t cc = (count - off < chunk) ? count - off : chunk;
    const int ln = (lanes > 1) ? turn % lanes : 0;
    hipStream_t st = (lanes > 1) ? c->lane[ln] : c->stream;
    double *X = (double *)c->ws + (size_t)ln * chunk * per_ct, *Y = X + cc * Xw;
    const u64 *pa = a + off * 2 * L * N, *pb = b + off * 2 * L * N;
    u64 *po = out + off * (relin ? 2 : 3) * L * N;
    if (n13) {  // four data limbs, eight blocks (reached with relin only: the plain multiply of this ring is bmul_big's)
      hipLaunchKernelGGL((k_bmul_front<13, 3, 4, 4>), dim3((unsigned)(cc * 4 * 16)), dim3(512), lds, st, c->dc, pa, pb, X);
      hipLaunchKernelGGL((k_bmul_mid<13, 10>), dim3((unsigned)(cc * nlm * 8)), dim3(256), (size_t)(4 * lds_words(10)) * 8, st, c->dc, (const double *)X, Y, nlm, L);
      hipLaunchKernelGGL((k_bmul_back<13, 3, 4, 4>), dim3((unsigned)(cc * 3 * 16)), dim3(512), lds, st, c->dc, (const double *)Y, po, X, 0);
      ABC_HIP_CHECK(hipGetLastError());
      if (bsplit_back13(c, st, cc, L, (const double *)X, Y, c->d_relin, po, 2 * (size_t)L * N, 1, po)) return 1;
      continue;
    }
    hipLaunchKernelGGL((k_bmul_front<14, 4, 8, 8>), dim3((unsigned)(cc * 4 * 32)), dim3(512), lds, st, c->dc, pa, pb, X);
    hipLaunchKernelGGL((k_bmul_mid<14, 10>), dim3((unsigned)(cc * nlm * 16)), dim3(256), (size_t)(4 * lds_words(10)) * 8, st, c->dc, (const double *)X, Y, nlm, L);
    hipLaunchKernelGGL((k_bmul_back<14, 4, 8, 8>), dim3((unsigned)(cc * 3 * 32)), dim3(512), lds, st, c->dc, (const double *)Y, po, X, relin ? 0 : 1);
    ABC_HIP_CHECK(hipGetLastError());
    if (relin && bsplit_back14(c, st, cc, L, (const double *)X, Y, c->d_relin, po, 2 * (size_t)L * N, 1, po)) return 1;
  }
  return scope.join();
}

}  // namespace abc
