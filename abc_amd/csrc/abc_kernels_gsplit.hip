// abc_kernels_gsplit.hip -- the split key switch without any LDS-resident limb: N = 2^15 (a limb is 256 KiB, LDS is 160), and
// the first step of the N = 2^14 sequence when too few ciphertexts are in flight to give every CU one of its 139 KiB workgroups.
//
// A 2^LOGN-point transform = NB = 2^(LOGN-10) blocks of 1024 points + a radix-NB "cross" pass over the NB values at one
// position of every block.  Forward: cross pass (registers), then block tails; inverse: block tails, then cross pass.
// So the whole CKKS key switch (and the multiply's front end) becomes
//   G1a k_gsplit_inv_tails  (limb j, block; four ciphertexts per workgroup, one wavefront each): operand block (a1 b1 for a
//                           multiply; the operand with the Galois gather folded in for a rotation) -> stages LOGN-1..LB of the
//                           inverse transform in 8.5 KiB of LDS, inverse twiddles of the block in a shared LDS table -> hinv
//   G1b k_gsplit_cross      (registers only): inverse cross pass, N^-1, canonical coefficient; per other key prime the forward
//                           cross pass -> half-done decomposition limbs `part`
//   G2a k_gsplit_special, G2b k_gsplit_pass, G2c k_gsplit_main: as k_split_special_fp (special prime) / k_split3_pass_fp /
//                           k_split4_main_fp of abc_kernels_fused.hip with NB blocks and radix-NB cross passes
// Same arithmetic as the N = 2^14 kernels (exact fp64 residues, primes < 2^50), bit-identical results
// (tests/test_gpu_configs.py, tests/test_gpu_paths.py).  Replaces, for N = 2^15, the generic sequence (expand / strided
// transform / inner product / tmod / finish kernels of abc_kernels_eval.hip: ~15 launches per rotation).
//   SealCiphertext::multiply (src/runtime/SealCiphertext.cpp:102-107), rotateRows (:52-61), relinearize -> gsplit_chunk
#include "abc_context.hpp"

namespace abc {

__device__ __forceinline__ double g_mulmod(double x, double y, double q, double qinv) {
  const double h = x * y;
  const double l = __builtin_fma(x, y, -h);
  const double c = __builtin_rint(h * qinv);
  return __builtin_fma(-c, q, h) + l;
}

// ---- G1a ----
// MODE 0: multiply (a, b: [ct][2][nl][N], operand = a1 b1).  MODE 1: operand in NTT form at a + ct * a_stride (Galois gather).
template <int LOGN, int MODE, bool GAL>
__global__ __launch_bounds__(256) void k_gsplit_inv_tails(DevCtx c, const u64 *__restrict__ a, const u64 *__restrict__ b,
                                                          size_t a_stride, double *__restrict__ hinv, int nl, int cc, u32 gelt) {
  constexpr int LOGNB = LOGN - 10, NB = 1 << LOGNB;
  extern __shared__ double dyn[];  // 4 transform buffers, then the block's inverse-twiddle table
  const int W = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int blk = blockIdx.x & (NB - 1);
  const int j = (int)((blockIdx.x >> LOGNB) % (unsigned)nl);
  const size_t ct = (size_t)((blockIdx.x >> LOGNB) / (unsigned)nl) * 4 + (size_t)W;
  const size_t N = (size_t)1 << LOGN, base = (size_t)blk << 10, PS = (size_t)c.ps, pw = (size_t)nl * N;
  const Mod m = mod_at(c, j);
  const FpTable t = fp_table(c, j);
  f64x2 *litw = reinterpret_cast<f64x2 *>(dyn + 4 * lds_words(10));
  f64x2 twv[4];
  block_twiddles_fetch<10, f64x2, 4>(t.itw, LOGNB, blk, (int)threadIdx.x, 256, twv);
  const bool live = ct < (size_t)cc;  // wavefront-uniform
  double x[16];
  if (live) {  // PassIdx<10, 8, 2>: slot 4g + k = element 4 (lane + 64 g) + k
    if (MODE == 0) {
      const u64 *__restrict__ a1 = a + ct * 2 * pw + pw + (size_t)j * N + base, *__restrict__ b1 = b + ct * 2 * pw + pw + (size_t)j * N + base;
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const u64 *pa = a1 + 4 * (lane + 64 * g), *pb = b1 + 4 * (lane + 64 * g);
        const u64x2 a0v = reinterpret_cast<const u64x2 *>(pa)[0], a1v = reinterpret_cast<const u64x2 *>(pa)[1];
        const u64x2 b0v = reinterpret_cast<const u64x2 *>(pb)[0], b1v = reinterpret_cast<const u64x2 *>(pb)[1];
        x[4 * g + 0] = g_mulmod(fp_from_u64(a0v.x), fp_from_u64(b0v.x), m.qd, m.qinv);
        x[4 * g + 1] = g_mulmod(fp_from_u64(a0v.y), fp_from_u64(b0v.y), m.qd, m.qinv);
        x[4 * g + 2] = g_mulmod(fp_from_u64(a1v.x), fp_from_u64(b1v.x), m.qd, m.qinv);
        x[4 * g + 3] = g_mulmod(fp_from_u64(a1v.y), fp_from_u64(b1v.y), m.qd, m.qinv);
      }
    } else {
      const u64 *__restrict__ sp = a + ct * a_stride + (size_t)j * N;
#pragma unroll
      for (int g = 0; g < 4; g++)
#pragma unroll
        for (int k = 0; k < 4; k++)
          x[4 * g + k] = fp_from_u64(sp[galois_ntt_src<GAL>((u32)(base + 4 * (lane + 64 * g) + k), gelt, LOGN)]);
    }
  }
  block_twiddles_store<10, f64x2, 4>(litw, (int)threadIdx.x, 256, twv);
  __syncthreads();
  if (live) {
    double *buf = dyn + W * lds_words(10);
    double *__restrict__ dst = hinv + (ct * nl + j) * PS + base;
    auto ld = [&](int r, int) { return x[r]; };
    auto st = [&](int, int i, double v) { dst[i] = v; };
    ntt_inv_block_a<10, FpArith, decltype(ld), decltype(st), true>(buf, ld, st, t, m, LOGNB, blk, lane, litw);
  }
}

// ---- radix-32 / 64 cross passes as two register passes with a transposition through an LDS tile [2^R blocks][32 positions] ----
// (N = 2^15 / 2^16; 256 threads).  In registers a radix-32 pass is 32 values per thread and the kernels around it held two such sets
// (k_gsplit_cross<15>: 256 VGPRs, one wavefront per SIMD; k_gsplit_pass<15>: 184).  Here a thread never holds more than eight:
//   inverse: stages R-1..RB on eight consecutive blocks (thread (group g, position p)), tile, stages RB-1..0 on the 2^RB blocks
//            j, j + 8, ... (thread (j, p)) -- which is where the values stay;
//   forward: stages 0..RB-1 from that same register layout, tile, stages RB..R-1 on eight consecutive blocks, store.
// The forward half rewrites exactly the tile words its thread read last, so one tile serves a whole kernel with two barriers per pass.
template <int R>
struct CrossLds {
  static constexpr int RA = 3, RB = R - RA, P = 32, NB = 1 << R;
  template <class Load>
  __device__ __forceinline__ static void inverse(double *lds, int tid, Load load, const FpTable &t, const FpK &kk, double (&x)[1 << RB]) {
    const int p = tid & (P - 1), jg = tid >> 5;
    if (tid < (1 << RB) * P) {
      double v[1 << RA];
#pragma unroll
      for (int j = 0; j < (1 << RA); j++) v[j] = fp_centre(load((jg << RA) + j, p), kk.q, kk.qinv);
#pragma unroll
      for (int u = R - 1; u >= RB; u--) {
        const int hf = 1 << (R - 1 - u);
#pragma unroll
        for (int j = 0; j < (1 << RA); j++) {
          if (j & hf) continue;
          FpArith::inv(v[j], v[j | hf], tw_load(t.itw + (1 << u) + (((jg << RA) + j) >> (R - u))), kk);
        }
      }
#pragma unroll
      for (int j = 0; j < (1 << RA); j++) lds[((jg << RA) + j) * P + p] = fp_centre(v[j], kk.q, kk.qinv);
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < (1 << RB); h++) x[h] = lds[((h << RA) + jg) * P + p];
#pragma unroll
    for (int u = RB - 1; u >= 0; u--) {
      const int hf = 1 << (RB - 1 - u);
#pragma unroll
      for (int h = 0; h < (1 << RB); h++) {
        if (h & hf) continue;
        FpArith::inv(x[h], x[h | hf], tw_load(t.itw + (1 << u) + (h >> (RB - u))), kk);
      }
    }
  }
  // x: values of blocks j + 8 h at position p, thread (j = tid >> 5, p = tid & 31); store(block, position, value)
  template <class Store>
  __device__ __forceinline__ static void forward(double *lds, int tid, const double (&x)[1 << RB], double add, const FpTable &t, const FpK &kk,
                                                 Store store) {
    const int p = tid & (P - 1), jg = tid >> 5;
    double y[1 << RB];
#pragma unroll
    for (int h = 0; h < (1 << RB); h++) y[h] = x[h] + add;
#pragma unroll
    for (int u = 0; u < RB; u++) {
      const int hf = 1 << (RB - 1 - u);
#pragma unroll
      for (int h = 0; h < (1 << RB); h++) {
        if (h & hf) continue;
        FpArith::fwd(y[h], y[h | hf], tw_load(t.tw + (1 << u) + (h >> (RB - u))), kk);
      }
    }
#pragma unroll
    for (int h = 0; h < (1 << RB); h++) lds[((h << RA) + jg) * P + p] = y[h];
    __syncthreads();
    if (tid < (1 << RB) * P) {
      double v[1 << RA];
#pragma unroll
      for (int j = 0; j < (1 << RA); j++) v[j] = lds[((jg << RA) + j) * P + p];
#pragma unroll
      for (int u = RB; u < R; u++) {
        const int hf = 1 << (R - 1 - u);
#pragma unroll
        for (int j = 0; j < (1 << RA); j++) {
          if (j & hf) continue;
          FpArith::fwd(v[j], v[j | hf], tw_load(t.tw + (1 << u) + (((jg << RA) + j) >> (R - u))), kk);
        }
      }
#pragma unroll
      for (int j = 0; j < (1 << RA); j++) store((jg << RA) + j, p, v[j]);
    }
    __syncthreads();  // the next pass rewrites the tile
  }
};

// ---- G1b ----
// LOGN = 14: registers only, grid (ct, j, quarter of the positions).  LOGN > 14: through CrossLds, grid (ct, j, group of 32 positions).
template <int LOGN>
__global__ __launch_bounds__(256) void k_gsplit_cross(DevCtx c, const double *__restrict__ hinv, double *__restrict__ part, int nl, int pack) {
  constexpr int LOGNB = LOGN - 10, NB = 1 << LOGNB;
  if constexpr (LOGN > 14) {
    using X = CrossLds<LOGNB>;
    __shared__ double lds[NB * X::P];
    const int pg = blockIdx.x & 31, tid = threadIdx.x;
    const int j = (int)((blockIdx.x >> 5) % (unsigned)nl);
    const size_t ct = (size_t)((blockIdx.x >> 5) / (unsigned)nl);
    const size_t PS = (size_t)c.ps;
    double x[1 << X::RB];
    {
      const Mod m = mod_at(c, j);
      const FpTable t = fp_table(c, j);
      const FpK kk = FpArith::consts(m);
      const double *__restrict__ src = hinv + (ct * nl + j) * PS + (size_t)(pg * X::P);
      X::inverse(lds, tid, [&](int k, int p) { return src[((size_t)k << 10) + p]; }, t, kk, x);
#pragma unroll
      for (int h = 0; h < (1 << X::RB); h++) {  // canonical [0, q_j) as a double
        const double w = fp_centre(fp_mul_lazy(x[h], m.inv_n_c, m.inv_n_cq, m.qd), m.qd, m.qinv);
        x[h] = w < 0.0 ? w + m.qd : w;
      }
    }
    for (int I = 0; I <= nl; I++) {
      if (I == j) continue;  // workgroup-uniform
      const int ki = (I == nl) ? c.K - 1 : I;
      const Mod m = mod_at(c, ki);
      const FpTable t = fp_table(c, ki);
      const FpK kk = FpArith::consts(m);
      double *__restrict__ dst = part + ((ct * (nl + 1) + I) * nl + j) * PS + (size_t)(pg * X::P);
      X::forward(lds, tid, x, 0.0, t, kk, [&](int k, int p, double v) { dst[((size_t)k << 10) + p] = v; });
    }
    return;
  }
  const int p = (int)((blockIdx.x & 3) << 8) + (int)threadIdx.x;
  const int j = (int)((blockIdx.x >> 2) % (unsigned)nl);
  const size_t ct = (size_t)((blockIdx.x >> 2) / (unsigned)nl);
  const size_t PS = (size_t)c.ps;
  double x[NB];
  {
    const Mod m = mod_at(c, j);
    const FpTable t = fp_table(c, j);
    const FpK kk = FpArith::consts(m);
    const double *__restrict__ src = hinv + (ct * nl + j) * PS;
#pragma unroll
    for (int k = 0; k < NB; k++) x[k] = fp_centre(src[(k << 10) + p], kk.q, kk.qinv);
    inv_cross<LOGNB>(x, t, kk);
#pragma unroll
    for (int k = 0; k < NB; k++) {  // canonical [0, q_j) as a double: the value SEAL's decomposition reduces modulo the other primes
      const double w = fp_centre(fp_mul_lazy(x[k], m.inv_n_c, m.inv_n_cq, m.qd), m.qd, m.qinv);
      x[k] = w < 0.0 ? w + m.qd : w;
    }
  }
  for (int I = 0; I <= nl; I++) {
    if (I == j) continue;
    const int ki = (I == nl) ? c.K - 1 : I;
    const Mod m = mod_at(c, ki);
    const FpTable t = fp_table(c, ki);
    const FpK kk = FpArith::consts(m);
    double y[NB];
#pragma unroll
    for (int k = 0; k < NB; k++) y[k] = x[k];
    fwd_cross<LOGNB>(y, t, kk);
    double *__restrict__ dst = part + ((ct * (nl + 1) + I) * nl + j) * PS;
    const int kind = (pack && I < nl) ? pack_kind(m.bits) : 0;  // packed half-done limbs (abc_ntt.hpp): N = 2^14, data primes only
    if (kind == 0) {
#pragma unroll
      for (int k = 0; k < NB; k++) dst[(k << 10) + p] = y[k];
    } else if (kind == 1) {
#pragma unroll
      for (int k = 0; k < NB; k++) pack_store<1>(dst, (size_t)1 << LOGN, (size_t)(k << 10) + p, fp_centre(y[k], kk.q, kk.qinv));
    } else {
#pragma unroll
      for (int k = 0; k < NB; k++) pack_store<2>(dst, (size_t)1 << LOGN, (size_t)(k << 10) + p, fp_centre(y[k], kk.q, kk.qinv));
    }
  }
}

// ---- G2a: special prime (cf. k_split_special_fp) ----
// ALL = false: the special prime only (CKKS: the data primes go through k_gsplit_main); output [ct][comp] limbs.
// ALL = true (BFV, operand in coefficient form: no diagonal term, no NTT-form output): every key prime I = 0..nl, grid
//   (ct, I, block); the inverse-transform tails of BOTH the special limb and the accumulated data limbs; output
//   [ct][I][comp] limbs -- k_bsplit_tcoef / k_bsplit_finish_big do the rest.
template <int LOGN, int NL, bool ALL>
__global__ __launch_bounds__(NL * 64, (ALL && NL == 8) ? 4 : 1) void k_gsplit_special(DevCtx c, const double *__restrict__ part, const u64 *__restrict__ key,
                                                            const double *__restrict__ keyf /* the key's fp64 twin, or null */,
                                                            double *__restrict__ tsp_half, int cc) {
  constexpr int LOGNB = LOGN - 10, NB = 1 << LOGNB, nl = NL;
  extern __shared__ double dyn[];
  const int J = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  // ALL: workgroup order (group of 8 key slices, ciphertext, slice within the group).  A key slice = (key prime I, block):
  // 2 nl KiB-blocks of the key that every ciphertext multiplies with.  blockIdx mod 8 picks the XCD, so one slice always meets
  // the same L2, and the cc workgroups that use it follow each other there: the slice is fetched once per call instead of once
  // per ciphertext (with (ct, I, block) order 18 slices of 128 KiB per XCD competed with the streaming operands at L = 8, and
  // 40 % of the key reads went past L2: tools/pmc_bfv.sh).
  unsigned slice, ctu;
  if (ALL) {
    const unsigned sx = blockIdx.x & 7, rest = blockIdx.x >> 3;
    ctu = rest % (unsigned)cc;
    slice = (rest / (unsigned)cc) * 8 + sx;
  } else {
    slice = (unsigned)nl * NB + (blockIdx.x & (NB - 1));
    ctu = blockIdx.x >> LOGNB;
  }
  const int blk = (int)(slice & (NB - 1));
  const int I = (int)(slice >> LOGNB);
  const size_t ct = (size_t)ctu;
  const size_t N = (size_t)1 << LOGN, base = (size_t)blk << 10, PS = (size_t)c.ps;
  const int ki = (I == nl) ? c.K - 1 : I;
  const Mod m = mod_at(c, ki);
  const FpTable t = fp_table(c, ki);
  const double q = m.qd, qinv = m.qinv;
  // NL = 8 (512 threads: one coefficient pair per thread): the key words of that pair are requested BEFORE the transform, so their
  // latency runs under it instead of being exposed after the barrier (the transform's own per-lane twiddle loads queue up behind
  // them on the in-order vector-memory counter, which costs nothing: all of it is one burst at the start)
  constexpr bool PREFETCH = ALL && NL == 8;
  u64x2 pk0[PREFETCH ? NL : 1], pk1[PREFETCH ? NL : 1];
  // key words: 16 raw bytes per (digit, component) either way -- the fp64 twin's words ARE the (centred) doubles (workgroup-uniform)
  const u64 *__restrict__ kw = keyf ? reinterpret_cast<const u64 *>(keyf) : key;
  auto kd = [&](u64 w) { return keyf ? __longlong_as_double((long long)w) : fp_from_u64(w); };
  if constexpr (PREFETCH) {
    const int e = 2 * (int)threadIdx.x;
#pragma unroll
    for (int Jx = 0; Jx < NL; Jx++) {
      pk0[Jx] = *reinterpret_cast<const u64x2 *>(kw + (((size_t)Jx * 2 + 0) * c.K + ki) * N + base + e);
      pk1[Jx] = *reinterpret_cast<const u64x2 *>(kw + (((size_t)Jx * 2 + 1) * c.K + ki) * N + base + e);
    }
  }
  {
    double *buf = dyn + J * lds_words(10);
    const double *__restrict__ src = part + ((ct * (nl + 1) + I) * nl + J) * PS + base;
    ntt_fwd_block_a<10, FpTail>(
        buf, [&](int, int i) { return fp_centre(src[i], q, qinv); }, [&](int, int i, double v) { buf[lds_pad(i)] = v; }, t, m, LOGNB,
        blk, lane);
  }
  __syncthreads();
  for (int e = 2 * (int)threadIdx.x; e < 1024; e += 2 * (int)blockDim.x) {
    double s0[2] = {0.0, 0.0}, s1[2] = {0.0, 0.0};
#pragma unroll
    for (int Jx = 0; Jx < NL; Jx++) {
      const f64x2 v = *reinterpret_cast<const f64x2 *>(dyn + Jx * lds_words(10) + lds_pad(e));
      u64x2 k0, k1;
      if constexpr (PREFETCH) {
        k0 = pk0[Jx];
        k1 = pk1[Jx];
      } else {
        k0 = *reinterpret_cast<const u64x2 *>(kw + (((size_t)Jx * 2 + 0) * c.K + ki) * N + base + e);
        k1 = *reinterpret_cast<const u64x2 *>(kw + (((size_t)Jx * 2 + 1) * c.K + ki) * N + base + e);
      }
      s0[0] += g_mulmod(v.x, kd(k0.x), q, qinv);
      s0[1] += g_mulmod(v.y, kd(k0.y), q, qinv);
      s1[0] += g_mulmod(v.x, kd(k1.x), q, qinv);
      s1[1] += g_mulmod(v.y, kd(k1.y), q, qinv);
      if (NL > 8 && (Jx & 7) == 7) {  // eight products of magnitude < q stay below 2^53; re-centre before adding more
        s0[0] = fp_centre(s0[0], q, qinv); s0[1] = fp_centre(s0[1], q, qinv);
        s1[0] = fp_centre(s1[0], q, qinv); s1[1] = fp_centre(s1[1], q, qinv);
      }
    }
    f64x2 r;
    r.x = fp_centre(s0[0], q, qinv); r.y = fp_centre(s0[1], q, qinv);
    *reinterpret_cast<f64x2 *>(dyn + lds_pad(e)) = r;
    r.x = fp_centre(s1[0], q, qinv); r.y = fp_centre(s1[1], q, qinv);
    *reinterpret_cast<f64x2 *>(dyn + lds_words(10) + lds_pad(e)) = r;
  }
  __syncthreads();
  for (int comp = J; comp < 2; comp += nl) {
    double *buf = dyn + comp * lds_words(10);
    double *__restrict__ dst = tsp_half + (ALL ? (ct * (nl + 1) + I) * 2 + comp : ct * 2 + comp) * PS + base;
    ntt_inv_block_a<10, FpArith>(
        buf, [&](int, int i) { return buf[lds_pad(i)]; }, [&](int, int i, double v) { dst[i] = v; }, t, m, LOGNB, blk, lane);
  }
}

// ---- G2a for eight digits in TWO rounds of four (BFV, ALL key primes): 256 threads, four transform buffers (35 KiB: four workgroups
// per CU where the 512-thread form above has two), the sums carried in registers across the rounds, two coefficient pairs per thread.
// Same arithmetic, same buffers as k_gsplit_special<LOGN, 8, true>.
template <int LOGN>
__global__ __launch_bounds__(256, 4) void k_bsplit_special8x2(DevCtx c, const double *__restrict__ part, const u64 *__restrict__ key,
                                                               const double *__restrict__ keyf /* the key's fp64 twin, or null */,
                                                               double *__restrict__ tsp_half, int cc) {
  constexpr int LOGNB = LOGN - 10, NB = 1 << LOGNB, nl = 8;
  __shared__ double dyn[4 * lds_words(10)];
  const int W = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const unsigned sx = blockIdx.x & 7, rest = blockIdx.x >> 3;  // workgroup order as above: a key slice stays with one XCD
  const unsigned ctu = rest % (unsigned)cc, slice = (rest / (unsigned)cc) * 8 + sx;
  const int blk = (int)(slice & (NB - 1));
  const int I = (int)(slice >> LOGNB);
  const size_t ct = (size_t)ctu;
  const size_t N = (size_t)1 << LOGN, base = (size_t)blk << 10, PS = (size_t)c.ps;
  const int ki = (I == nl) ? c.K - 1 : I;
  const Mod m = mod_at(c, ki);
  const FpTable t = fp_table(c, ki);
  const double q = m.qd, qinv = m.qinv;
  double s0[2][2] = {{0.0, 0.0}, {0.0, 0.0}}, s1[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
  double *buf = dyn + W * lds_words(10);
#pragma nounroll
  for (int r = 0; r < 2; r++) {
    const int J = 4 * r + W;
    const double *__restrict__ src = part + ((ct * (nl + 1) + I) * nl + J) * PS + base;
    double xin[16];
#pragma unroll
    for (int k = 0; k < 16; k++) xin[k] = src[(k << 6) + lane];
    if (r) __syncthreads();  // round 0's buffers have been consumed
    ntt_fwd_block_a<10, FpTail>(
        buf, [&](int s, int) { return fp_centre(xin[s], q, qinv); }, [&](int, int i, double v) { buf[lds_pad(i)] = v; }, t, m, LOGNB, blk,
        lane);
    __syncthreads();
#pragma unroll
    for (int pp = 0; pp < 2; pp++) {
      const int e = 2 * (int)threadIdx.x + 512 * pp;
#pragma unroll
      for (int Jx = 0; Jx < 4; Jx++) {
        const f64x2 v = *reinterpret_cast<const f64x2 *>(dyn + Jx * lds_words(10) + lds_pad(e));
        const size_t w0 = (((size_t)(4 * r + Jx) * 2 + 0) * c.K + ki) * N + base + e, w1 = w0 + (size_t)c.K * N;
        f64x2 y0, y1;
        if (keyf) {  // workgroup-uniform: the words ARE the (centred) doubles, no conversion
          y0 = *reinterpret_cast<const f64x2 *>(keyf + w0);
          y1 = *reinterpret_cast<const f64x2 *>(keyf + w1);
        } else {
          const u64x2 k0 = *reinterpret_cast<const u64x2 *>(key + w0), k1 = *reinterpret_cast<const u64x2 *>(key + w1);
          y0.x = fp_from_u64(k0.x); y0.y = fp_from_u64(k0.y);
          y1.x = fp_from_u64(k1.x); y1.y = fp_from_u64(k1.y);
        }
        s0[pp][0] += g_mulmod(v.x, y0.x, q, qinv);
        s0[pp][1] += g_mulmod(v.y, y0.y, q, qinv);
        s1[pp][0] += g_mulmod(v.x, y1.x, q, qinv);
        s1[pp][1] += g_mulmod(v.y, y1.y, q, qinv);
      }
    }
  }
  // park the centred sums in buffers 0 and 1: a thread rewrites only words it alone read in the last round
#pragma unroll
  for (int pp = 0; pp < 2; pp++) {
    const int e = 2 * (int)threadIdx.x + 512 * pp;
    f64x2 rr;
    rr.x = fp_centre(s0[pp][0], q, qinv); rr.y = fp_centre(s0[pp][1], q, qinv);
    *reinterpret_cast<f64x2 *>(dyn + lds_pad(e)) = rr;
    rr.x = fp_centre(s1[pp][0], q, qinv); rr.y = fp_centre(s1[pp][1], q, qinv);
    *reinterpret_cast<f64x2 *>(dyn + lds_words(10) + lds_pad(e)) = rr;
  }
  __syncthreads();
  if (W < 2) {
    double *b2 = dyn + W * lds_words(10);
    double *__restrict__ dst = tsp_half + ((ct * (nl + 1) + I) * 2 + W) * PS + base;
    ntt_inv_block_a<10, FpArith>(
        b2, [&](int, int i) { return b2[lds_pad(i)]; }, [&](int, int i, double v) { dst[i] = v; }, t, m, LOGNB, blk, lane);
  }
}

// ---- G2b: special-prime limb back to coefficients (cross pass), + q_sp/2; forward cross pass of (t mod q_j + fix) per data prime ----
template <int LOGN>
__global__ __launch_bounds__(256) void k_gsplit_pass(DevCtx c, const double *__restrict__ tsp_half, double *__restrict__ tpart, int nl) {
  constexpr int LOGNB = LOGN - 10, NB = 1 << LOGNB;
  if constexpr (LOGN > 14) {  // through CrossLds, grid ((ct, comp), group of 32 positions)
    using X = CrossLds<LOGNB>;
    __shared__ double lds[NB * X::P];
    const int pg = blockIdx.x & 31, tid = threadIdx.x;
    const size_t cc = blockIdx.x >> 5;
    const size_t PS = (size_t)c.ps;
    double x[1 << X::RB];
    {
      const Mod ms = mod_at(c, c.K - 1);
      const FpTable ts = fp_table(c, c.K - 1);
      const FpK ks = FpArith::consts(ms);
      const double *__restrict__ src = tsp_half + cc * PS + (size_t)(pg * X::P);
      X::inverse(lds, tid, [&](int k, int p) { return src[((size_t)k << 10) + p]; }, ts, ks, x);
      const double half = (double)(ms.q >> 1);
#pragma unroll
      for (int h = 0; h < (1 << X::RB); h++) {
        const double w = fp_centre(fp_mul_lazy(x[h], ms.inv_n_c, ms.inv_n_cq, ms.qd) + half, ms.qd, ms.qinv);
        x[h] = w < 0.0 ? w + ms.qd : w;
      }
    }
    const u64 halfq = c.mods[c.K - 1].q >> 1;
    for (int j = 0; j < nl; j++) {
      const Mod m = mod_at(c, j);
      const FpTable t = fp_table(c, j);
      const FpK kk = FpArith::consts(m);
      const u64 hm = reduce64(halfq, m);
      const double fix = hm ? (double)(m.q - hm) : 0.0;
      double *__restrict__ dst = tpart + (cc * nl + j) * PS + (size_t)(pg * X::P);
      X::forward(lds, tid, x, fix, t, kk, [&](int k, int p, double v) { dst[((size_t)k << 10) + p] = v; });
    }
    return;
  }
  const size_t cc = blockIdx.x >> 2;
  const int p = (int)((blockIdx.x & 3) << 8) + (int)threadIdx.x;
  const size_t PS = (size_t)c.ps;
  double x[NB];
  {
    const Mod ms = mod_at(c, c.K - 1);
    const FpTable ts = fp_table(c, c.K - 1);
    const FpK ks = FpArith::consts(ms);
    const double *__restrict__ src = tsp_half + cc * PS;
#pragma unroll
    for (int k = 0; k < NB; k++) x[k] = fp_centre(src[(k << 10) + p], ks.q, ks.qinv);
    inv_cross<LOGNB>(x, ts, ks);
    const double half = (double)(ms.q >> 1);
#pragma unroll
    for (int k = 0; k < NB; k++) {
      const double w = fp_centre(fp_mul_lazy(x[k], ms.inv_n_c, ms.inv_n_cq, ms.qd) + half, ms.qd, ms.qinv);
      x[k] = w < 0.0 ? w + ms.qd : w;
    }
  }
  const u64 half = c.mods[c.K - 1].q >> 1;
  for (int j = 0; j < nl; j++) {
    const Mod m = mod_at(c, j);
    const FpTable t = fp_table(c, j);
    const FpK kk = FpArith::consts(m);
    const u64 hm = reduce64(half, m);
    const double fix = hm ? (double)(m.q - hm) : 0.0;
    double y[NB];
#pragma unroll
    for (int k = 0; k < NB; k++) y[k] = x[k] + fix;
    fwd_cross<LOGNB>(y, t, kk);
    double *__restrict__ dst = tpart + (cc * nl + j) * PS;
#pragma unroll
    for (int k = 0; k < NB; k++) dst[(k << 10) + p] = y[k];
  }
}

// ---- G2c: main (cf. k_split4_main_fp) ----
template <int MODE, int NL>
struct GPairOps {
  u64x2 k0[NL], k1[NL];
  u64x2 a0, a1, b0, b1;
  u64 xs[2], d0s[2], d1s[2];
};

template <int LOGN, int MODE, bool GAL, int NL>
__global__ __launch_bounds__(512, NL <= 4 ? 4 : 2) void k_gsplit_main(DevCtx c, const double *__restrict__ part, const double *__restrict__ tpart,
                                                        const u64 *__restrict__ opa, const u64 *__restrict__ opb, size_t opa_stride,
                                                        size_t opb_stride, int add_c1, const u64 *__restrict__ key, u64 *__restrict__ out,
                                                        u32 gelt, u32 imap, int ni) {
  // grid (ct, slot, block), slot < ni; the data prime of a slot is nibble `slot` of imap (all: 0x76543210, ni = nl; a subset for
  // chains that mix fp64-capable and wider primes: abc_kernels_isplit.hip)
  constexpr int LOGNB = LOGN - 10, NB = 1 << LOGNB, nl = NL, NT = 512, PER = 2;
  extern __shared__ double dyn[];
  static_assert(NL + 1 <= 8, "one wavefront per limb, eight wavefronts");
  const int W = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int blk = blockIdx.x & (NB - 1);
  const int I = (int)((imap >> (4 * ((blockIdx.x >> LOGNB) % (unsigned)ni))) & 15u);
  const size_t ct = (size_t)((blockIdx.x >> LOGNB) / (unsigned)ni);
  const size_t N = (size_t)1 << LOGN, base = (size_t)blk << 10, PS = (size_t)c.ps;
  const Mod m = mod_at(c, I);
  const FpTable t = fp_table(c, I);
  const double q = m.qd, qinv = m.qinv;
  f64x2 *ltw = reinterpret_cast<f64x2 *>(dyn + (nl + 1) * lds_words(10));
  const size_t pw = (size_t)nl * N;
  const ABC_CONST_AS DevConst *cst = (const ABC_CONST_AS DevConst *)c.cst;
  const double inv = cst->inv_special_c[I], inv_q = cst->inv_special_cq[I];

  f64x2 twv[PER];
  block_twiddles_fetch<10, f64x2, PER>(t.tw, LOGNB, blk, (int)threadIdx.x, NT, twv);
  const bool has_limb = W <= nl;
  const int Wc = has_limb ? W : 0;
  const double *__restrict__ src = (Wc < nl - 1) ? part + ((ct * (nl + 1) + I) * nl + (Wc < I ? Wc : Wc + 1)) * PS + base
                                                 : tpart + ((ct * 2 + (Wc - (nl - 1))) * nl + I) * PS + base;
  double xin[16];
  if (has_limb) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const f64x2 v = *reinterpret_cast<const f64x2 *>(src + (k << 7) + 2 * lane);
      xin[k] = v.x;
      xin[8 + k] = v.y;
    }
  }
  const int e = 2 * (int)threadIdx.x;
  GPairOps<MODE, NL> o;
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
      const u32 si = galois_ntt_src<GAL>((u32)(base + e + k), gelt, LOGN);
      o.xs[k] = xl[si];
      o.d0s[k] = o.d1s[k] = 0;
      if (opb) {
        const u64 *ad = opb + ct * opb_stride + (size_t)I * N;
        o.d0s[k] = ad[si];
        if (add_c1) o.d1s[k] = ad[pw + si];
      }
    }
  }
  block_twiddles_store<10, f64x2, PER>(ltw, (int)threadIdx.x, NT, twv);
  __syncthreads();
  if (has_limb) {
    double *buf = dyn + W * lds_words(10);
#pragma unroll
    for (int r = 0; r < 16; r++) xin[r] = fp_centre(xin[r], q, qinv);
    ntt_fwd_tail1024_pairs<FpTail>(buf, xin, [&](int, int i, double v) { buf[lds_pad(i)] = v; }, t, m, LOGNB, blk, lane, ltw);
  }
  __syncthreads();
  const double *tt0 = dyn + (nl - 1) * lds_words(10), *tt1 = dyn + nl * lds_words(10);
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
          x[k] = g_mulmod(x1[k], y1[k], q, qinv);
          d0[k] = g_mulmod(x0[k], y0[k], q, qinv);
          d1[k] = g_mulmod(x0[k], y1[k], q, qinv) + g_mulmod(x1[k], y0[k], q, qinv);
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
    s0[0] += g_mulmod(x[0], fp_from_u64(o.k0[Jx].x), q, qinv);
    s0[1] += g_mulmod(x[1], fp_from_u64(o.k0[Jx].y), q, qinv);
    s1[0] += g_mulmod(x[0], fp_from_u64(o.k1[Jx].x), q, qinv);
    s1[1] += g_mulmod(x[1], fp_from_u64(o.k1[Jx].y), q, qinv);
  }
  const f64x2 u0 = *reinterpret_cast<const f64x2 *>(tt0 + lds_pad(e)), u1 = *reinterpret_cast<const f64x2 *>(tt1 + lds_pad(e));
  u64x2 r;
  r.x = fp_to_canon(fp_mul_lazy(s0[0] - u0.x, inv, inv_q, q) + d0[0], q, qinv);
  r.y = fp_to_canon(fp_mul_lazy(s0[1] - u0.y, inv, inv_q, q) + d0[1], q, qinv);
  *reinterpret_cast<u64x2 *>(out + ((ct * 2 + 0) * nl + I) * N + base + e) = r;
  r.x = fp_to_canon(fp_mul_lazy(s1[0] - u1.x, inv, inv_q, q) + d1[0], q, qinv);
  r.y = fp_to_canon(fp_mul_lazy(s1[1] - u1.y, inv, inv_q, q) + d1[1], q, qinv);
  *reinterpret_cast<u64x2 *>(out + ((ct * 2 + 1) * nl + I) * N + base + e) = r;
}

// ---- G2c for deep chains (8 to 15 data limbs): 1024 threads, one wavefront per limb (nl - 1 decomposition limbs + the two
// mod-down limbs), one workgroup per CU (up to 152 KiB of LDS), key words loaded inside the sum instead of ahead of the transform
// (16 x 16 bytes per limb would not fit the registers).  Same arithmetic, same buffers as k_gsplit_main.
template <int LOGN, int MODE, bool GAL>
__global__ __launch_bounds__(1024) void k_gsplit_main_deep(DevCtx c, const double *__restrict__ part, const double *__restrict__ tpart,
                                                           const u64 *__restrict__ opa, const u64 *__restrict__ opb, size_t opa_stride,
                                                           size_t opb_stride, int add_c1, const u64 *__restrict__ key, u64 *__restrict__ out,
                                                           u32 gelt, int nl, u64 imap, int ni) {
  // grid (ct, slot, block), slot < ni; the data prime of a slot is nibble `slot` of imap (all: 0xfedcba9876543210, ni = nl)
  constexpr int LOGNB = LOGN - 10, NB = 1 << LOGNB, NT = 1024;
  extern __shared__ double dyn[];
  const int W = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int blk = blockIdx.x & (NB - 1);
  const int I = (int)((imap >> (4 * ((blockIdx.x >> LOGNB) % (unsigned)ni))) & 15u);
  const size_t ct = (size_t)((blockIdx.x >> LOGNB) / (unsigned)ni);
  const size_t N = (size_t)1 << LOGN, base = (size_t)blk << 10, PS = (size_t)c.ps;
  const Mod m = mod_at(c, I);
  const FpTable t = fp_table(c, I);
  const double q = m.qd, qinv = m.qinv;
  f64x2 *ltw = reinterpret_cast<f64x2 *>(dyn + (nl + 1) * lds_words(10));
  const size_t pw = (size_t)nl * N;
  const ABC_CONST_AS DevConst *cst = (const ABC_CONST_AS DevConst *)c.cst;
  const double inv = cst->inv_special_c[I], inv_q = cst->inv_special_cq[I];
  f64x2 twv[1];
  block_twiddles_fetch<10, f64x2, 1>(t.tw, LOGNB, blk, (int)threadIdx.x, NT, twv);
  const bool has_limb = W <= nl;
  const int Wc = has_limb ? W : 0;
  const double *__restrict__ src = (Wc < nl - 1) ? part + ((ct * (nl + 1) + I) * nl + (Wc < I ? Wc : Wc + 1)) * PS + base
                                                 : tpart + ((ct * 2 + (Wc - (nl - 1))) * nl + I) * PS + base;
  double xin[16];
  if (has_limb) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const f64x2 v = *reinterpret_cast<const f64x2 *>(src + (k << 7) + 2 * lane);
      xin[k] = v.x;
      xin[8 + k] = v.y;
    }
  }
  block_twiddles_store<10, f64x2, 1>(ltw, (int)threadIdx.x, NT, twv);
  __syncthreads();
  if (has_limb) {
    double *buf = dyn + W * lds_words(10);
    if (m.bits >= 49) {
#pragma unroll
      for (int r = 0; r < 16; r++) xin[r] = fp_centre(xin[r], q, qinv);
    }
    ntt_fwd_tail1024_pairs<FpArith>(buf, xin, [&](int, int i, double v) { buf[lds_pad(i)] = v; }, t, m, LOGNB, blk, lane, ltw);
  }
  __syncthreads();
  if (threadIdx.x >= 512) return;  // one coefficient pair per thread of the first eight wavefronts
  const int e = 2 * (int)threadIdx.x;
  const double *tt0 = dyn + (nl - 1) * lds_words(10), *tt1 = dyn + nl * lds_words(10);
  double s0[2] = {0.0, 0.0}, s1[2] = {0.0, 0.0}, d0[2] = {0.0, 0.0}, d1[2] = {0.0, 0.0};
  for (int Jx = 0; Jx < nl; Jx++) {
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
          x[k] = g_mulmod(x1[k], y1[k], q, qinv);
          d0[k] = g_mulmod(x0[k], y0[k], q, qinv);
          d1[k] = g_mulmod(x0[k], y1[k], q, qinv) + g_mulmod(x1[k], y0[k], q, qinv);
        }
      } else {
        const u64 *xl = opa + ct * opa_stride + (size_t)I * N;
#pragma unroll
        for (int k = 0; k < 2; k++) {
          const u32 si = galois_ntt_src<GAL>((u32)(base + e + k), gelt, LOGN);
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
    s0[0] += g_mulmod(x[0], fp_from_u64(k0.x), q, qinv);
    s0[1] += g_mulmod(x[1], fp_from_u64(k0.y), q, qinv);
    s1[0] += g_mulmod(x[0], fp_from_u64(k1.x), q, qinv);
    s1[1] += g_mulmod(x[1], fp_from_u64(k1.y), q, qinv);
    if ((Jx & 7) == 7) {
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
}

// ---- host side ----
// first step only (the half-done decomposition limbs): used by the N = 2^14 sequence for small batches
template <int LOGN>
static void launch_gsplit_front(hipStream_t st, abc_hip_ctx *c, size_t cc, int nl, int mode, const u64 *opa, const u64 *opb,
                                size_t opa_stride, double *hinv, double *part, u32 gelt, int pack = 0) {
  constexpr int NB = 1 << (LOGN - 10);
  const dim3 g1((unsigned)(((cc + 3) / 4) * nl * NB)), g2((unsigned)(cc * nl * (LOGN > 14 ? 32 : 4)));
  const size_t lds = (size_t)(4 * lds_words(10)) * 8 + 1024 * 16;
  if (mode == 0)
    hipLaunchKernelGGL((k_gsplit_inv_tails<LOGN, 0, false>), g1, dim3(256), lds, st, c->dc, opa, opb, 0, hinv, nl, (int)cc, 0u);
  else if (gelt)
    hipLaunchKernelGGL((k_gsplit_inv_tails<LOGN, 1, true>), g1, dim3(256), lds, st, c->dc, opa, nullptr, opa_stride, hinv, nl, (int)cc, gelt);
  else
    hipLaunchKernelGGL((k_gsplit_inv_tails<LOGN, 1, false>), g1, dim3(256), lds, st, c->dc, opa, nullptr, opa_stride, hinv, nl, (int)cc, 0u);
  hipLaunchKernelGGL((k_gsplit_cross<LOGN>), g2, dim3(256), 0, st, c->dc, hinv, part, nl, pack);
}
void gsplit_front14(hipStream_t st, abc_hip_ctx *c, size_t cc, int nl, int mode, const u64 *opa, const u64 *opb, size_t opa_stride,
                    double *hinv, double *part, u32 gelt, int pack) {
  launch_gsplit_front<14>(st, c, cc, nl, mode, opa, opb, opa_stride, hinv, part, gelt, pack);
}

template <int LOGN>
static void launch_gsplit_back(hipStream_t st, abc_hip_ctx *c, size_t cc, int nl, int mode, const u64 *opa, const u64 *opb,
                               size_t opa_stride, size_t opb_stride, int add_c1, const u64 *key, double *part, double *tpart,
                               double *tsp, u64 *out, u32 gelt) {
  constexpr int NB = 1 << (LOGN - 10);
  const size_t lds_sp = (size_t)((nl < 2 ? 2 : nl) * lds_words(10)) * 8;
  const size_t lds_main = (size_t)((nl + 1) * lds_words(10)) * 8 + 1024 * 16;
  const dim3 gsp((unsigned)(cc * NB)), gmain((unsigned)(cc * nl * NB));
  if (nl > 7) {  // deep chains: one wavefront per limb still, but up to sixteen of them
#define ABC_GSPD(NLV) hipLaunchKernelGGL((k_gsplit_special<LOGN, NLV, false>), gsp, dim3(64 * NLV), lds_sp, st, c->dc, part, key, key_twin_lookup(c, key), tsp, (int)cc)
    switch (nl) {
      case 8: ABC_GSPD(8); break;
      case 9: ABC_GSPD(9); break;
      case 10: ABC_GSPD(10); break;
      case 11: ABC_GSPD(11); break;
      case 12: ABC_GSPD(12); break;
      case 13: ABC_GSPD(13); break;
      case 14: ABC_GSPD(14); break;
      default: ABC_GSPD(15); break;
    }
#undef ABC_GSPD
    hipLaunchKernelGGL((k_gsplit_pass<LOGN>), dim3((unsigned)(cc * 2 * (LOGN > 14 ? 32 : 4))), dim3(256), 0, st, c->dc, tsp, tpart, nl);
    if (mode == 0)
      hipLaunchKernelGGL((k_gsplit_main_deep<LOGN, 0, false>), gmain, dim3(1024), lds_main, st, c->dc, part, tpart, opa, opb, opa_stride,
                         opb_stride, add_c1, key, out, gelt, nl, 0xfedcba9876543210ull, nl);
    else if (gelt)
      hipLaunchKernelGGL((k_gsplit_main_deep<LOGN, 1, true>), gmain, dim3(1024), lds_main, st, c->dc, part, tpart, opa, opb, opa_stride,
                         opb_stride, add_c1, key, out, gelt, nl, 0xfedcba9876543210ull, nl);
    else
      hipLaunchKernelGGL((k_gsplit_main_deep<LOGN, 1, false>), gmain, dim3(1024), lds_main, st, c->dc, part, tpart, opa, opb, opa_stride,
                         opb_stride, add_c1, key, out, gelt, nl, 0xfedcba9876543210ull, nl);
    return;
  }
#define ABC_GSP(NLV)                                                                                                                    \
  hipLaunchKernelGGL((k_gsplit_special<LOGN, NLV, false>), gsp, dim3(64 * NLV), lds_sp, st, c->dc, part, key, key_twin_lookup(c, key), tsp, (int)cc);                            \
  hipLaunchKernelGGL((k_gsplit_pass<LOGN>), dim3((unsigned)(cc * 2 * (LOGN > 14 ? 32 : 4))), dim3(256), 0, st, c->dc, tsp, tpart, nl);  \
  if (mode == 0)                                                                                                                        \
    hipLaunchKernelGGL((k_gsplit_main<LOGN, 0, false, NLV>), gmain, dim3(512), lds_main, st, c->dc, part, tpart, opa, opb, opa_stride,  \
                       opb_stride, add_c1, key, out, gelt, 0x76543210u, nl);                                                                             \
  else if (gelt)                                                                                                                        \
    hipLaunchKernelGGL((k_gsplit_main<LOGN, 1, true, NLV>), gmain, dim3(512), lds_main, st, c->dc, part, tpart, opa, opb, opa_stride,   \
                       opb_stride, add_c1, key, out, gelt, 0x76543210u, nl);                                                                             \
  else                                                                                                                                  \
    hipLaunchKernelGGL((k_gsplit_main<LOGN, 1, false, NLV>), gmain, dim3(512), lds_main, st, c->dc, part, tpart, opa, opb, opa_stride,  \
                       opb_stride, add_c1, key, out, gelt, 0x76543210u, nl)
  switch (nl) {
    case 1: ABC_GSP(1); break;
    case 2: ABC_GSP(2); break;
    case 3: ABC_GSP(3); break;
    case 4: ABC_GSP(4); break;
    case 5: ABC_GSP(5); break;
    case 6: ABC_GSP(6); break;
    default: ABC_GSP(7); break;
  }
#undef ABC_GSP
}

// the deep-chain main step of N = 2^15 over a subset of the data primes (mixed chains of 8 to 15 limbs, abc_kernels_isplit.hip)
void gsplit_main_deep_subset15(hipStream_t st, abc_hip_ctx *c, size_t cc, int nl, int mode, const double *part, const double *tpart,
                               const u64 *opa, const u64 *opb, size_t opa_stride, size_t opb_stride, int add_c1, const u64 *key, u64 *out,
                               u32 gelt, u64 imap, int ni) {
  if (ni < 1) return;
  const size_t lds_main = (size_t)((nl + 1) * lds_words(10)) * 8 + 1024 * 16;
  const dim3 gmain((unsigned)(cc * ni * 32));
  if (mode == 0)
    hipLaunchKernelGGL((k_gsplit_main_deep<15, 0, false>), gmain, dim3(1024), lds_main, st, c->dc, part, tpart, opa, opb, opa_stride, opb_stride,
                       add_c1, key, out, gelt, nl, imap, ni);
  else if (gelt)
    hipLaunchKernelGGL((k_gsplit_main_deep<15, 1, true>), gmain, dim3(1024), lds_main, st, c->dc, part, tpart, opa, opb, opa_stride, opb_stride,
                       add_c1, key, out, gelt, nl, imap, ni);
  else
    hipLaunchKernelGGL((k_gsplit_main_deep<15, 1, false>), gmain, dim3(1024), lds_main, st, c->dc, part, tpart, opa, opb, opa_stride, opb_stride,
                       add_c1, key, out, gelt, nl, imap, ni);
}

// the main step of N = 2^15 over a subset of the data primes (mixed chains, abc_kernels_isplit.hip): mode 0 multiply, mode 1 key switch
bool gsplit_main_subset15(hipStream_t st, abc_hip_ctx *c, size_t cc, int nl, int mode, const double *part, const double *tpart, const u64 *opa,
                          const u64 *opb, size_t opa_stride, size_t opb_stride, int add_c1, const u64 *key, u64 *out, u32 gelt, u32 imap,
                          int ni) {
  if (nl < 1 || nl > 7 || ni < 1) return ni == 0;
  const size_t lds_main = (size_t)((nl + 1) * lds_words(10)) * 8 + 1024 * 16;
  const dim3 gmain((unsigned)(cc * ni * 32));
#define ABC_GSUB(NLV)                                                                                                                  \
  if (mode == 0)                                                                                                                       \
    hipLaunchKernelGGL((k_gsplit_main<15, 0, false, NLV>), gmain, dim3(512), lds_main, st, c->dc, part, tpart, opa, opb, opa_stride,   \
                       opb_stride, add_c1, key, out, gelt, imap, ni);                                                                  \
  else if (gelt)                                                                                                                       \
    hipLaunchKernelGGL((k_gsplit_main<15, 1, true, NLV>), gmain, dim3(512), lds_main, st, c->dc, part, tpart, opa, opb, opa_stride,    \
                       opb_stride, add_c1, key, out, gelt, imap, ni);                                                                  \
  else                                                                                                                                 \
    hipLaunchKernelGGL((k_gsplit_main<15, 1, false, NLV>), gmain, dim3(512), lds_main, st, c->dc, part, tpart, opa, opb, opa_stride,   \
                       opb_stride, add_c1, key, out, gelt, imap, ni)
  switch (nl) {
    case 1: ABC_GSUB(1); break;
    case 2: ABC_GSUB(2); break;
    case 3: ABC_GSUB(3); break;
    case 4: ABC_GSUB(4); break;
    case 5: ABC_GSUB(5); break;
    case 6: ABC_GSUB(6); break;
    default: ABC_GSUB(7); break;
  }
#undef ABC_GSUB
  return true;
}

// scratch (words, limb stride c->dc.ps): hinv nl | part nl(nl+1) | tpart 2 nl | tsp_half 2
size_t gsplit_scratch_words(const abc_hip_ctx *c, int nl) {
  return ((size_t)nl + (size_t)nl * (nl + 1) + 2 * (size_t)nl + 2) * (size_t)c->dc.ps;
}
bool gsplit_applies(const abc_hip_ctx *c, int nl) {
  if (c->logn != 15 || c->scheme != 2 || !c->use_fp || c->sw.no_gsplit || nl < 1 || nl > 15) return false;
  for (int j = 0; j < c->K; j++)
    if (!fp_ok(c->h_mods[j].bits)) return false;
  return true;
}
// one chunk at N = 2^15: mode 0 multiply (opa = a, opb = b), mode 1 key switch (opa = operand in NTT form, opb = addend)
int gsplit_chunk15(abc_hip_ctx *c, hipStream_t st, u64 *scratch, size_t cc, int nl, int mode, const u64 *opa, const u64 *opb,
                   size_t opa_stride, size_t opb_stride, int add_c1, const u64 *key, u64 *out, u32 gelt) {
  const size_t PS = (size_t)c->dc.ps;
  double *hinv = (double *)scratch, *part = hinv + cc * (size_t)nl * PS, *tpart = part + cc * (size_t)nl * (nl + 1) * PS,
         *tsp = tpart + cc * 2 * (size_t)nl * PS;
  launch_gsplit_front<15>(st, c, cc, nl, mode, opa, opb, opa_stride, hinv, part, gelt);
  launch_gsplit_back<15>(st, c, cc, nl, mode, opa, opb, opa_stride, opb_stride, add_c1, key, part, tpart, tsp, out, gelt);
  ABC_HIP_CHECK(hipGetLastError());
  return 0;
}

// ---- BFV (coefficient-form ciphertexts), N = 2^14: the key switch behind k_fused_operand_pass0_fp<14, false, false> ----
// k_gsplit_special<14, NL, true> leaves, per (ct, key prime, component), the accumulated limb after stages 13..4 of its inverse
// transform.  k_bsplit_tcoef / k_bsplit_finish_big (defined with the N = 2^15 / 2^16 sequence below; registers only) finish the
// special limb once per (ct, component) and each data limb: cross pass, N^-1, (x - (t mod q_I + fix)) q_sp^-1 (+ addend) in
// coefficient form -- k_fused_ks_moddown_bfv_fp without its 139 KiB workgroup.  (A single kernel that recomputed the special
// limb's cross pass per data prime from registers was 1 - 3 % slower.)
template <int LOGN>
__global__ void k_bsplit_tcoef(DevCtx c, const double *__restrict__ half, double *__restrict__ tco, int nl);
template <int LOGN>
__global__ void k_bsplit_finish_big(DevCtx c, const double *__restrict__ half, const double *__restrict__ tco, const u64 *__restrict__ addend,
                                    size_t addend_stride, int add_c1, u64 *__restrict__ out, int nl, u32 ginv);

// half: [cc][nl+1][2] limbs at stride c->dc.ps; part as written by k_fused_operand_pass0_fp<14, false, false> (padded layout)
bool bsplit_applies(const abc_hip_ctx *c, int nl) {
  if (c->logn != 14 || c->scheme != 1 || !c->use_fp || c->sw.no_bsplit || nl < 1 || nl > 8) return false;
  for (int j = 0; j < c->K; j++)
    if (!fp_ok(c->h_mods[j].bits)) return false;
  return true;
}
int bsplit_back14(abc_hip_ctx *c, hipStream_t st, size_t cc, int nl, const double *part, double *half, const u64 *key, const u64 *addend,
                  size_t addend_stride, int add_c1, u64 *out, u32 ginv) {
  const dim3 g((unsigned)(cc * (nl + 1) * 16));
  const size_t lds = (size_t)((nl < 2 ? 2 : nl) * lds_words(10)) * 8;
  if (nl == 8 && !c->sw.no_special8x2) {  // two rounds of four digits, four workgroups per CU: +2 % multiply, +5 % rotate
    hipLaunchKernelGGL((k_bsplit_special8x2<14>), g, dim3(256), 0, st, c->dc, part, key, key_twin_lookup(c, key), half, (int)cc);
  } else
#define ABC_BSP(NLV) hipLaunchKernelGGL((k_gsplit_special<14, NLV, true>), g, dim3(64 * NLV), lds, st, c->dc, part, key, key_twin_lookup(c, key), half, (int)cc)
  switch (nl) {
    case 1: ABC_BSP(1); break;
    case 2: ABC_BSP(2); break;
    case 3: ABC_BSP(3); break;
    case 4: ABC_BSP(4); break;
    case 5: ABC_BSP(5); break;
    case 6: ABC_BSP(6); break;
    case 7: ABC_BSP(7); break;
    default: ABC_BSP(8); break;
  }
#undef ABC_BSP
  // half occupies the ksacc + tsp regions of the caller's scratch (2 nl + 2 limbs per ciphertext), tco the tlast region behind them
  double *tco = half + cc * 2 * (size_t)(nl + 1) * (size_t)c->dc.ps;
  hipLaunchKernelGGL((k_bsplit_tcoef<14>), dim3((unsigned)(cc * 2 * 4)), dim3(256), 0, st, c->dc, half, tco, nl);
  hipLaunchKernelGGL((k_bsplit_finish_big<14>), dim3((unsigned)(cc * 2 * nl * 4)), dim3(256), 0, st, c->dc, half, tco, addend,
                     addend_stride, add_c1, out, nl, ginv);
  ABC_HIP_CHECK(hipGetLastError());
  return 0;
}

// the same three launches behind a `part` of eight blocks (N = 2^13: abc_kernels_bmul.hip writes it behind the floor, as at 2^14)
int bsplit_back13(abc_hip_ctx *c, hipStream_t st, size_t cc, int nl, const double *part, double *half, const u64 *key, const u64 *addend,
                  size_t addend_stride, int add_c1, u64 *out, u32 ginv) {
  const dim3 g((unsigned)(cc * (nl + 1) * 8));
  const size_t lds = (size_t)((nl < 2 ? 2 : nl) * lds_words(10)) * 8;
#define ABC_BSP(NLV) hipLaunchKernelGGL((k_gsplit_special<13, NLV, true>), g, dim3(64 * NLV), lds, st, c->dc, part, key, key_twin_lookup(c, key), half, (int)cc)
  switch (nl) {
    case 1: ABC_BSP(1); break;
    case 2: ABC_BSP(2); break;
    case 3: ABC_BSP(3); break;
    case 4: ABC_BSP(4); break;
    case 5: ABC_BSP(5); break;
    case 6: ABC_BSP(6); break;
    case 7: ABC_BSP(7); break;
    default: ABC_BSP(8); break;
  }
#undef ABC_BSP
  double *tco = half + cc * 2 * (size_t)(nl + 1) * (size_t)c->dc.ps;
  hipLaunchKernelGGL((k_bsplit_tcoef<13>), dim3((unsigned)(cc * 2 * 4)), dim3(256), 0, st, c->dc, half, tco, nl);
  hipLaunchKernelGGL((k_bsplit_finish_big<13>), dim3((unsigned)(cc * 2 * nl * 4)), dim3(256), 0, st, c->dc, half, tco, addend,
                     addend_stride, add_c1, out, nl, ginv);
  ABC_HIP_CHECK(hipGetLastError());
  return 0;
}

// ---- BFV, N = 2^15 / 2^16 (coefficient-form ciphertexts, every key prime below 2^50): the same three steps with NB = 32 / 64 ----
// B1 k_bsplit_pass0   (ct, J, key prime I, quarter): digit J of the operand, read as it lies (no reduction modulo q_I is needed in
//                     fp64), forward cross pass modulo q_I -> half-done limb (ct, I, J)
// B2 k_gsplit_special<LOGN, NL, true>: tails + inner product + inverse tails, every key prime
// B3 k_bsplit_tcoef   : special limb back to canonical coefficients (+ q_sp/2) -- once per (ct, comp), where the N = 2^14 finish
//                     kernel recomputes it per data prime from registers (16 values; here it would be 64)
// B4 k_bsplit_finish_big: inverse cross pass of data limb I, N^-1, subtract, q_sp^-1, addend -> coefficient form
// Replaces, for these rings, k_ks_expand_strided_fp + block transforms + k_ks_inner + transforms + k_ks_tmod + k_ks_finish.
template <int LOGN>
__global__ __launch_bounds__(256) void k_bsplit_pass0(DevCtx c, const u64 *__restrict__ src, size_t src_stride, double *__restrict__ part,
                                                      int nl, u32 ginv = 0 /* BFV rotation: elt^-1 mod 2N, the permutation folded into the load */) {
  constexpr int LOGNB = LOGN - 10, NB = 1 << LOGNB;
  const int p = (int)((blockIdx.x & 3) << 8) + (int)threadIdx.x;
  const unsigned w = blockIdx.x >> 2;
  const int I = (int)(w % (unsigned)(nl + 1));
  const int J = (int)((w / (unsigned)(nl + 1)) % (unsigned)nl);
  const size_t ct = (size_t)(w / (unsigned)(nl + 1) / (unsigned)nl);
  const size_t N = (size_t)1 << LOGN, PS = (size_t)c.ps;
  const int ki = (I == nl) ? c.K - 1 : I;
  const Mod m = mod_at(c, ki);
  const FpTable t = fp_table(c, ki);
  const FpK kk = FpArith::consts(m);
  const u64 *__restrict__ sp = src + ct * src_stride + (size_t)J * N;
  double x[NB];
  if (ginv) {  // workgroup-uniform
    const u64 qj = c.mods[J].q;
#pragma unroll
    for (int k = 0; k < NB; k++) {
      bool neg;
      const u64 v = sp[galois_coef_src((u32)((k << 10) + p), ginv, LOGN, neg)];
      x[k] = fp_from_u64(neg ? neg_mod(v, qj) : v);
    }
  } else {
#pragma unroll
    for (int k = 0; k < NB; k++) x[k] = fp_from_u64(sp[(k << 10) + p]);
  }
  fwd_cross<LOGNB>(x, t, kk);
  double *__restrict__ dst = part + ((ct * (nl + 1) + I) * nl + J) * PS;
#pragma unroll
  for (int k = 0; k < NB; k++) dst[(k << 10) + p] = x[k];
}

template <int LOGN>
__global__ __launch_bounds__(256) void k_bsplit_tcoef(DevCtx c, const double *__restrict__ half, double *__restrict__ tco, int nl) {
  constexpr int LOGNB = LOGN - 10, NB = 1 << LOGNB;
  const int p = (int)((blockIdx.x & 3) << 8) + (int)threadIdx.x;
  const size_t cc = blockIdx.x >> 2;  // ct*2 + comp
  const size_t ct = cc >> 1;
  const int comp = (int)(cc & 1);
  const size_t PS = (size_t)c.ps;
  const Mod ms = mod_at(c, c.K - 1);
  const FpTable ts = fp_table(c, c.K - 1);
  const FpK ks = FpArith::consts(ms);
  const double *__restrict__ src = half + ((ct * (nl + 1) + nl) * 2 + comp) * PS;
  double t[NB];
#pragma unroll
  for (int k = 0; k < NB; k++) t[k] = fp_centre(src[(k << 10) + p], ks.q, ks.qinv);
  inv_cross<LOGNB>(t, ts, ks);
  const double hq = (double)(ms.q >> 1);
  double *__restrict__ dst = tco + cc * PS;
#pragma unroll
  for (int k = 0; k < NB; k++) {
    const double w = fp_centre(fp_mul_lazy(t[k], ms.inv_n_c, ms.inv_n_cq, ms.qd) + hq, ms.qd, ms.qinv);
    dst[(k << 10) + p] = w < 0.0 ? w + ms.qd : w;  // canonical [0, q_sp)
  }
}

template <int LOGN>
__global__ __launch_bounds__(256) void k_bsplit_finish_big(DevCtx c, const double *__restrict__ half, const double *__restrict__ tco,
                                                           const u64 *__restrict__ addend, size_t addend_stride, int add_c1,
                                                           u64 *__restrict__ out, int nl, u32 ginv /* BFV rotation: gather the addend */) {
  constexpr int LOGNB = LOGN - 10, NB = 1 << LOGNB;
  const int p = (int)((blockIdx.x & 3) << 8) + (int)threadIdx.x;
  const int I = (int)((blockIdx.x >> 2) % (unsigned)nl);
  const size_t cc = (size_t)((blockIdx.x >> 2) / (unsigned)nl);  // ct*2 + comp
  const size_t ct = cc >> 1;
  const int comp = (int)(cc & 1);
  const size_t N = (size_t)1 << LOGN, PS = (size_t)c.ps;
  const Mod m = mod_at(c, I);
  const FpTable tb = fp_table(c, I);
  const FpK kk = FpArith::consts(m);
  double x[NB];
  {
    const double *__restrict__ src = half + ((ct * (nl + 1) + I) * 2 + comp) * PS;
#pragma unroll
    for (int k = 0; k < NB; k++) x[k] = fp_centre(src[(k << 10) + p], kk.q, kk.qinv);
    inv_cross<LOGNB>(x, tb, kk);
  }
  const u64 hm = reduce64(c.mods[c.K - 1].q >> 1, m);
  const double fix = hm ? (double)(m.q - hm) : 0.0;
  const ABC_CONST_AS DevConst *cst = (const ABC_CONST_AS DevConst *)c.cst;
  const double inv = cst->inv_special_c[I], inv_q = cst->inv_special_cq[I];
  const double *__restrict__ tsrc = tco + cc * PS;
  u64 *__restrict__ o = out + (cc * nl + I) * N;
  const bool add = addend && (comp == 0 || add_c1);
  const u64 *__restrict__ cin = add ? addend + ct * addend_stride + ((size_t)comp * nl + I) * N : nullptr;
#pragma unroll
  for (int k = 0; k < NB; k++) {
    const double d = fp_mul_lazy(x[k], m.inv_n_c, m.inv_n_cq, m.qd) - (tsrc[(k << 10) + p] + fix);
    double r = fp_mul_lazy(d, inv, inv_q, m.qd);
    if (add) {
      if (ginv) {  // workgroup-uniform
        bool neg;
        const u64 v = cin[galois_coef_src((u32)((k << 10) + p), ginv, LOGN, neg)];
        r += fp_from_u64(neg ? neg_mod(v, m.q) : v);
      } else {
        r += fp_from_u64(cin[(k << 10) + p]);
      }
    }
    o[(k << 10) + p] = fp_to_canon(r, m.qd, m.qinv);
  }
}

// B1 through LDS for N = 2^15 / 2^16: workgroup (ct, digit J, group of 32 positions) reads the digit ONCE and runs the forward
// cross pass modulo every key prime I from registers -- stages 0..RB-1 on blocks 2^RA apart, transposition through 16 KiB of LDS,
// stages RB..R-1 on 2^RA consecutive blocks: at most eight values per thread (k_bsplit_pass0<16>: 64 values, 209 VGPRs, and one
// read of the digit per key prime: 72 limb reads per ciphertext where 8 do).
template <int LOGN>
__global__ __launch_bounds__(256) void k_bsplit_pass0_lds(DevCtx c, const u64 *__restrict__ src, size_t src_stride, double *__restrict__ part,
                                                          int nl, u32 ginv /* BFV rotation: elt^-1 mod 2N, the permutation folded into the load */) {
  constexpr int R = LOGN - 10, NB = 1 << R, RA = 3, RB = R - RA, P = 32;
  __shared__ double lds[NB * P];
  const int pg = blockIdx.x & 31;
  const int J = (int)((blockIdx.x >> 5) % (unsigned)nl);
  const size_t ct = (size_t)((blockIdx.x >> 5) / (unsigned)nl);
  const size_t N = (size_t)1 << LOGN, PS = (size_t)c.ps;
  const int tid = threadIdx.x;
  constexpr int JOBS1 = (1 << RA) * P, JOBS2 = (1 << RB) * P;  // 256 and 256 (R = 6) / 128 (R = 5)
  const bool w1 = tid < JOBS1, w2 = tid < JOBS2;
  const int p = tid & (P - 1), jg = tid >> 5;  // phase 1: j = jg (block index mod 2^RA); phase 2: g = jg (group of 2^RA blocks)
  const u64 *__restrict__ sp = src + ct * src_stride + (size_t)J * N + (size_t)(pg * P);
  double y0[1 << RB];
  if (w1 && ginv) {  // workgroup-uniform second condition
    const u64 *__restrict__ limb = src + ct * src_stride + (size_t)J * N;
    const u64 qj = c.mods[J].q;
#pragma unroll
    for (int h = 0; h < (1 << RB); h++) {
      bool neg;
      const u64 v = limb[galois_coef_src((u32)(((h << RA) + jg) * 1024 + pg * P + p), ginv, LOGN, neg)];
      y0[h] = fp_from_u64(neg ? neg_mod(v, qj) : v);
    }
  } else if (w1) {
#pragma unroll
    for (int h = 0; h < (1 << RB); h++) y0[h] = fp_from_u64(sp[(size_t)((h << RA) + jg) * 1024 + p]);
  }
  for (int I = 0; I <= nl; I++) {
    const int ki = (I == nl) ? c.K - 1 : I;
    const Mod m = mod_at(c, ki);
    const FpTable t = fp_table(c, ki);
    const FpK kk = FpArith::consts(m);
    if (w1) {
      double y[1 << RB];
#pragma unroll
      for (int h = 0; h < (1 << RB); h++) y[h] = y0[h];
#pragma unroll
      for (int u = 0; u < RB; u++) {
        const int hf = 1 << (RB - 1 - u);
#pragma unroll
        for (int h = 0; h < (1 << RB); h++) {
          if (h & hf) continue;
          FpArith::fwd(y[h], y[h | hf], tw_load(t.tw + (1 << u) + (h >> (RB - u))), kk);
        }
      }
#pragma unroll
      for (int h = 0; h < (1 << RB); h++) lds[((h << RA) + jg) * P + p] = y[h];
    }
    __syncthreads();
    if (w2) {
      double x[1 << RA];
#pragma unroll
      for (int j = 0; j < (1 << RA); j++) x[j] = lds[((jg << RA) + j) * P + p];
#pragma unroll
      for (int u = RB; u < R; u++) {
        const int hf = 1 << (R - 1 - u);
#pragma unroll
        for (int j = 0; j < (1 << RA); j++) {
          if (j & hf) continue;
          FpArith::fwd(x[j], x[j | hf], tw_load(t.tw + (1 << u) + (((jg << RA) + j) >> (R - u))), kk);
        }
      }
      double *__restrict__ dst = part + ((ct * (nl + 1) + I) * nl + J) * PS + (size_t)(pg * P) + p;
#pragma unroll
      for (int j = 0; j < (1 << RA); j++) dst[(size_t)((jg << RA) + j) * 1024] = x[j];
    }
    __syncthreads();  // the next prime's first pass rewrites the tile
  }
}

// B4 through LDS for N = 2^15 / 2^16: the radix-32 / 64 inverse cross pass as TWO register passes of at most eight values with a
// transposition in 16 KiB of LDS between them, instead of 32 / 64 values per thread (k_bsplit_finish_big<16>: 256 VGPRs, one
// wavefront per SIMD, 1.6 TB/s on its bytes).  Workgroup ((ct, comp), I, group of 32 positions); stages R-1..RB run on RA
// consecutive block indices (thread = (group of 2^RA blocks, position)), stages RB-1..0 on blocks 2^RA apart (thread = (block
// index mod 2^RA, position)), which then finishes its 2^RB coefficients: N^-1, subtract the special limb, q_sp^-1, addend.
template <int LOGN>
__global__ __launch_bounds__(256) void k_bsplit_finish_lds(DevCtx c, const double *__restrict__ half, const double *__restrict__ tco,
                                                           const u64 *__restrict__ addend, size_t addend_stride, int add_c1,
                                                           u64 *__restrict__ out, int nl, u32 ginv /* BFV rotation: gather the addend */) {
  constexpr int R = LOGN - 10, NB = 1 << R, RA = 3, RB = R - RA, P = 32;
  __shared__ double lds[NB * P];
  const int pg = blockIdx.x & 31;
  const int I = (int)((blockIdx.x >> 5) % (unsigned)nl);
  const size_t cc = (size_t)((blockIdx.x >> 5) / (unsigned)nl);  // ct*2 + comp
  const size_t ct = cc >> 1;
  const int comp = (int)(cc & 1);
  const size_t N = (size_t)1 << LOGN, PS = (size_t)c.ps;
  const Mod m = mod_at(c, I);
  const FpTable tb = fp_table(c, I);
  const FpK kk = FpArith::consts(m);
  const int tid = threadIdx.x;
  {
    const double *__restrict__ src = half + ((ct * (nl + 1) + I) * 2 + comp) * PS + (size_t)(pg * P);
    for (int job = tid; job < (1 << RB) * P; job += 256) {
      const int p = job & (P - 1), g = job >> 5;
      double x[1 << RA];
#pragma unroll
      for (int j = 0; j < (1 << RA); j++) x[j] = fp_centre(src[(size_t)((g << RA) + j) * 1024 + p], kk.q, kk.qinv);
#pragma unroll
      for (int u = R - 1; u >= RB; u--) {
        const int hf = 1 << (R - 1 - u);
#pragma unroll
        for (int j = 0; j < (1 << RA); j++) {
          if (j & hf) continue;
          FpArith::inv(x[j], x[j | hf], tw_load(tb.itw + (1 << u) + (((g << RA) + j) >> (R - u))), kk);
        }
      }
#pragma unroll
      for (int j = 0; j < (1 << RA); j++) lds[((g << RA) + j) * P + p] = fp_centre(x[j], kk.q, kk.qinv);
    }
  }
  __syncthreads();
  const u64 hm = reduce64(c.mods[c.K - 1].q >> 1, m);
  const double fix = hm ? (double)(m.q - hm) : 0.0;
  const ABC_CONST_AS DevConst *cst = (const ABC_CONST_AS DevConst *)c.cst;
  const double inv = cst->inv_special_c[I], inv_q = cst->inv_special_cq[I];
  const double *__restrict__ tsrc = tco + cc * PS + (size_t)(pg * P);
  u64 *__restrict__ o = out + (cc * nl + I) * N + (size_t)(pg * P);
  const bool add = addend && (comp == 0 || add_c1);
  const u64 *__restrict__ cin = add ? addend + ct * addend_stride + ((size_t)comp * nl + I) * N + (size_t)(pg * P) : nullptr;
  for (int job = tid; job < (1 << RA) * P; job += 256) {
    const int p = job & (P - 1), j = job >> 5;
    double y[1 << RB];
#pragma unroll
    for (int h = 0; h < (1 << RB); h++) y[h] = lds[((h << RA) + j) * P + p];
#pragma unroll
    for (int u = RB - 1; u >= 0; u--) {
      const int hf = 1 << (RB - 1 - u);
#pragma unroll
      for (int h = 0; h < (1 << RB); h++) {
        if (h & hf) continue;
        FpArith::inv(y[h], y[h | hf], tw_load(tb.itw + (1 << u) + (h >> (RB - u))), kk);
      }
    }
#pragma unroll
    for (int h = 0; h < (1 << RB); h++) {
      const size_t e = (size_t)((h << RA) + j) * 1024 + p;
      const double d = fp_mul_lazy(y[h], m.inv_n_c, m.inv_n_cq, m.qd) - (tsrc[e] + fix);
      double r = fp_mul_lazy(d, inv, inv_q, m.qd);
      if (add) {
        if (ginv) {  // workgroup-uniform
          bool neg;
          const u64 v = (cin - (size_t)(pg * P))[galois_coef_src((u32)(e + (size_t)(pg * P)), ginv, LOGN, neg)];
          r += fp_from_u64(neg ? neg_mod(v, m.q) : v);
        } else {
          r += fp_from_u64(cin[e]);
        }
      }
      o[e] = fp_to_canon(r, m.qd, m.qinv);
    }
  }
}

// N = 2^13 (BFVDefault(8192): eight 1024-point blocks, radix-8 cross passes in registers) takes the same sequence
bool bsplit_big_applies(const abc_hip_ctx *c, int nl) {
  if ((c->logn != 13 && c->logn != 15 && c->logn != 16) || c->scheme != 1 || !c->use_fp || c->sw.no_bsplit || c->sw.no_gsplit || nl < 1 || nl > 8) return false;
  for (int j = 0; j < c->K; j++)
    if (!fp_ok(c->h_mods[j].bits)) return false;
  return true;
}

template <int LOGN>
static int bsplit_big_chunk(abc_hip_ctx *c, hipStream_t st, double *scratch, size_t cc, int nl, const u64 *target, size_t target_stride,
                            const u64 *key, const u64 *addend, size_t addend_stride, int add_c1, u64 *out, u32 ginv = 0) {
  if (ginv && LOGN > 14 && c->sw.no_finish_lds) { set_error("bsplit: the folded BFV permutation needs the LDS cross passes on these rings"); return 1; }
  constexpr int NB = 1 << (LOGN - 10);
  const size_t PS = (size_t)c->dc.ps;
  double *part = scratch, *half = part + cc * (size_t)nl * (nl + 1) * PS, *tco = half + cc * 2 * (size_t)(nl + 1) * PS;
  constexpr bool REG = LOGN < 15;  // cross passes of at most 16 values stay in registers
  if constexpr (REG)
    hipLaunchKernelGGL((k_bsplit_pass0<LOGN>), dim3((unsigned)(cc * nl * (nl + 1) * 4)), dim3(256), 0, st, c->dc, target, target_stride, part,
                       nl, ginv);
  else if (c->sw.no_finish_lds)
    hipLaunchKernelGGL((k_bsplit_pass0<LOGN>), dim3((unsigned)(cc * nl * (nl + 1) * 4)), dim3(256), 0, st, c->dc, target, target_stride, part,
                       nl, 0u);
  else
    hipLaunchKernelGGL((k_bsplit_pass0_lds<LOGN>), dim3((unsigned)(cc * nl * 32)), dim3(256), 0, st, c->dc, target, target_stride, part, nl, ginv);
  const dim3 g((unsigned)(cc * (nl + 1) * NB));
  const size_t lds = (size_t)((nl < 2 ? 2 : nl) * lds_words(10)) * 8;
  if (nl == 8 && !c->sw.no_special8x2) {
    hipLaunchKernelGGL((k_bsplit_special8x2<LOGN>), g, dim3(256), 0, st, c->dc, part, key, key_twin_lookup(c, key), half, (int)cc);
  } else
#define ABC_BSPB(NLV) hipLaunchKernelGGL((k_gsplit_special<LOGN, NLV, true>), g, dim3(64 * NLV), lds, st, c->dc, part, key, key_twin_lookup(c, key), half, (int)cc)
  switch (nl) {
    case 1: ABC_BSPB(1); break;
    case 2: ABC_BSPB(2); break;
    case 3: ABC_BSPB(3); break;
    case 4: ABC_BSPB(4); break;
    case 5: ABC_BSPB(5); break;
    case 6: ABC_BSPB(6); break;
    case 7: ABC_BSPB(7); break;
    default: ABC_BSPB(8); break;
  }
#undef ABC_BSPB
  hipLaunchKernelGGL((k_bsplit_tcoef<LOGN>), dim3((unsigned)(cc * 2 * 4)), dim3(256), 0, st, c->dc, half, tco, nl);
  if constexpr (REG)
    hipLaunchKernelGGL((k_bsplit_finish_big<LOGN>), dim3((unsigned)(cc * 2 * nl * 4)), dim3(256), 0, st, c->dc, half, tco, addend,
                       addend_stride, add_c1, out, nl, ginv);
  else if (c->sw.no_finish_lds)
    hipLaunchKernelGGL((k_bsplit_finish_big<LOGN>), dim3((unsigned)(cc * 2 * nl * 4)), dim3(256), 0, st, c->dc, half, tco, addend,
                       addend_stride, add_c1, out, nl, 0u);
  else
    hipLaunchKernelGGL((k_bsplit_finish_lds<LOGN>), dim3((unsigned)(cc * 2 * nl * 32)), dim3(256), 0, st, c->dc, half, tco, addend,
                       addend_stride, add_c1, out, nl, ginv);
  ABC_HIP_CHECK(hipGetLastError());
  return 0;
}

// whole call: chunks of at most 1 GiB (N = 2^13) / 4 GiB of scratch on the context's stream (a chunk at these sizes fills the device by itself)
int bsplit_big(abc_hip_ctx *c, const u64 *target, size_t target_stride, const u64 *key, u64 *out, int nl, size_t count, const u64 *addend,
               size_t addend_stride, bool add_c1, u32 ginv) {
  const size_t N = (size_t)c->n, PS = (size_t)c->dc.ps;
  const size_t per_ct = ((size_t)nl * (nl + 1) + 2 * (size_t)(nl + 1) + 2) * PS;  // part | half | tco (words)
  // N = 2^15 / 2^16: 4 GiB of scratch, i.e. 85 ciphertexts per launch group at N = 2^16, L = 8 (1 GiB = 21: config 5 -3.5 % -- a key
  // slice is fetched once per launch group and XCD, k_bsplit_special8x2)
  size_t chunk = (((size_t)(c->logn > 14 ? 4 : 1) << 30) / 8) / per_ct;
  if (chunk < 1) chunk = 1;
  if (chunk > count) chunk = count;
  else if (count % chunk && count / chunk < 8) chunk = (count + count / chunk) / (count / chunk + 1);  // even chunks, no runt
  if (ensure_workspace(c, chunk * per_ct * 8)) return 1;
  (void)key_twin(c, key);  // the inner-product kernels read the key's fp64 twin where it exists
  for (size_t off = 0; off < count; off += chunk) {
    const size_t cc = (count - off < chunk) ? count - off : chunk;
    const u64 *tg = target + off * target_stride;
    const u64 *ad = addend ? addend + off * addend_stride : nullptr;
    u64 *o = out + off * 2 * (size_t)nl * N;
    if (c->logn == 13) {
      if (bsplit_big_chunk<13>(c, c->stream, (double *)c->ws, cc, nl, tg, target_stride, key, ad, addend_stride, add_c1 ? 1 : 0, o, ginv)) return 1;
      continue;
    }
    const int rc = (c->logn == 15)
                       ? bsplit_big_chunk<15>(c, c->stream, (double *)c->ws, cc, nl, tg, target_stride, key, ad, addend_stride, add_c1 ? 1 : 0, o, ginv)
                       : bsplit_big_chunk<16>(c, c->stream, (double *)c->ws, cc, nl, tg, target_stride, key, ad, addend_stride, add_c1 ? 1 : 0, o, ginv);
    if (rc) return rc;
  }
  return 0;
}

}  // namespace abc
