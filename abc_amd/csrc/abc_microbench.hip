// abc_microbench.hip -- issue-rate probes for the two arithmetic pipes the NTT can run on:
// the 32-bit integer multiplier (v_mad_u64_u32 / v_mul_hi_u32) and the fp64 FMA pipe.  They decide
// which modular-multiplication scheme the hot kernels use (DESIGN.md "Arithmetic roofline").
#include "abc_context.hpp"

// Probe kernels, not product: compiled only with -DABC_HIP_WITH_MICROBENCH (python -m abc_amd.build --microbench, which
// tools/microbench.py asks for).  The default library keeps the entry point and answers that it was built without them.
#ifdef ABC_HIP_WITH_MICROBENCH
namespace abc {

// 4 independent dependent-chains per lane so the result measures throughput, not latency.
__global__ __launch_bounds__(256) void k_mb_shoup(const Mod *mods, u64 *sink, int iters) {
  const Mod m = mods[0];
  u64 x0 = threadIdx.x + 1, x1 = x0 + 77, x2 = x0 + 1234, x3 = x0 + 99999;
  const u64 w = m.q / 3 + blockIdx.x, ws = mulhi64(w, m.mu);  // any constants
  for (int i = 0; i < iters; i++) {
    x0 = mul_shoup_lazy(x0, w, ws, m.q);
    x1 = mul_shoup_lazy(x1, w, ws, m.q);
    x2 = mul_shoup_lazy(x2, w, ws, m.q);
    x3 = mul_shoup_lazy(x3, w, ws, m.q);
  }
  sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3;
}

__global__ __launch_bounds__(256) void k_mb_barrett(const Mod *mods, u64 *sink, int iters) {
  const Mod m = mods[0];
  u64 x0 = threadIdx.x + 1, x1 = x0 + 77, x2 = x0 + 1234, x3 = x0 + 99999;
  const u64 w = m.q / 3 + blockIdx.x;
  for (int i = 0; i < iters; i++) {
    x0 = mul_mod(x0, w, m);
    x1 = mul_mod(x1, w, m);
    x2 = mul_mod(x2, w, m);
    x3 = mul_mod(x3, w, m);
  }
  sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3;
}

__device__ __forceinline__ double fp_mulmod(double a, double b, double q, double qinv) {
  const double h = a * b;
  const double l = __builtin_fma(a, b, -h);
  const double e = __builtin_floor(h * qinv);
  double r = __builtin_fma(-e, q, h) + l;
  r = r < 0.0 ? r + q : r;
  r = r >= q ? r - q : r;
  return r;
}

__global__ __launch_bounds__(256) void k_mb_fp64(const Mod *mods, u64 *sink, int iters) {
  const Mod m = mods[0];
  double x0 = threadIdx.x + 1, x1 = x0 + 77, x2 = x0 + 1234, x3 = x0 + 99999;
  const double w = (double)(m.q / 3 + blockIdx.x);
  for (int i = 0; i < iters; i++) {
    x0 = fp_mulmod(x0, w, m.qd, m.qinv);
    x1 = fp_mulmod(x1, w, m.qd, m.qinv);
    x2 = fp_mulmod(x2, w, m.qd, m.qinv);
    x3 = fp_mulmod(x3, w, m.qd, m.qinv);
  }
  sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = (u64)x0 ^ (u64)x1 ^ (u64)x2 ^ (u64)x3;
}

__global__ __launch_bounds__(256) void k_mb_mulwide(const Mod *mods, u64 *sink, int iters) {
  u64 x0 = threadIdx.x + 1, x1 = x0 + 77, x2 = x0 + 1234, x3 = x0 + 99999;
  const u64 w = mods[0].q / 3 + blockIdx.x;
  for (int i = 0; i < iters; i++) {
    U128 a = mul_wide(x0, w), b = mul_wide(x1, w), cc = mul_wide(x2, w), d = mul_wide(x3, w);
    x0 = a.lo ^ a.hi; x1 = b.lo ^ b.hi; x2 = cc.lo ^ cc.hi; x3 = d.lo ^ d.hi;
  }
  sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3;
}

__global__ __launch_bounds__(256) void k_mb_fma64(const Mod *mods, u64 *sink, int iters) {
  double x0 = threadIdx.x + 1, x1 = x0 + 77, x2 = x0 + 1234, x3 = x0 + 99999;
  const double w = 1.0 + 1e-9 * blockIdx.x, z = 1e-7;
  for (int i = 0; i < iters; i++) {
    x0 = __builtin_fma(x0, w, z); x1 = __builtin_fma(x1, w, z); x2 = __builtin_fma(x2, w, z); x3 = __builtin_fma(x3, w, z);
  }
  sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = (u64)(x0 + x1 + x2 + x3);
}

// which: 0 Shoup lazy modmul, 1 Barrett modmul, 2 fp64 modmul, 3 64x64->128 product, 4 fp64 FMA.
// Each lane performs 4*iters operations; grid = 256 CUs x 8 workgroups x 256 lanes.
int microbench_instr(abc_hip_ctx *c, int which, int iters, double *ms);
int microbench_bfly(abc_hip_ctx *c, int which, int iters, double *ms);
int microbench_ntt(abc_hip_ctx *c, int which, int iters, double *ms);
int microbench(abc_hip_ctx *c, int which, int iters, double *ms) {
  if (which >= 300) return microbench_ntt(c, which - 300, iters, ms);
  if (which >= 200) return microbench_bfly(c, which - 200, iters, ms);
  if (which >= 100) return microbench_instr(c, which - 100, iters, ms);
  const int blocks = 256 * 8, threads = 256;
  if (ensure_workspace(c, (size_t)blocks * threads * 8)) return 1;
  u64 *sink = (u64 *)c->ws;
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    ABC_HIP_CHECK(hipEventRecord(c->ev0, c->stream));
    switch (which) {
      case 0: hipLaunchKernelGGL(k_mb_shoup, dim3(blocks), dim3(threads), 0, c->stream, c->d_mods, sink, iters); break;
      case 1: hipLaunchKernelGGL(k_mb_barrett, dim3(blocks), dim3(threads), 0, c->stream, c->d_mods, sink, iters); break;
      case 2: hipLaunchKernelGGL(k_mb_fp64, dim3(blocks), dim3(threads), 0, c->stream, c->d_mods, sink, iters); break;
      case 3: hipLaunchKernelGGL(k_mb_mulwide, dim3(blocks), dim3(threads), 0, c->stream, c->d_mods, sink, iters); break;
      case 4: hipLaunchKernelGGL(k_mb_fma64, dim3(blocks), dim3(threads), 0, c->stream, c->d_mods, sink, iters); break;
      default: set_error("microbench: unknown probe"); return 1;
    }
    ABC_HIP_CHECK(hipGetLastError());
    ABC_HIP_CHECK(hipEventRecord(c->ev1, c->stream));
    ABC_HIP_CHECK(hipEventSynchronize(c->ev1));
    float t = 0;
    ABC_HIP_CHECK(hipEventElapsedTime(&t, c->ev0, c->ev1));
    if (t < best) best = t;
  }
  *ms = best;
  return 0;
}

}  // namespace abc

// ---- raw instruction issue-rate probes (inline asm, 8 independent instructions per iteration) ----
namespace abc {
#define ABC_ASM8(INS)                                                                                                   \
  asm volatile(INS "\n" INS "\n" INS "\n" INS "\n" INS "\n" INS "\n" INS "\n" INS "\n" ::: "v10", "v11", "v12", "v13", "vcc")

template <int WHICH>
__global__ __launch_bounds__(256) void k_mb_instr(u64 *sink, int iters) {
  u32 a = threadIdx.x | 1, b = blockIdx.x | 3;
  asm volatile("v_mov_b32 v10, %0\n v_mov_b32 v11, %1\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n" ::"v"(a), "v"(b) : "v10", "v11", "v12", "v13");
  for (int i = 0; i < iters; i++) {
    if (WHICH == 0) ABC_ASM8("v_mad_u64_u32 v[12:13], vcc, v10, v11, v[12:13]");
    if (WHICH == 1) ABC_ASM8("v_mul_lo_u32 v12, v10, v11");
    if (WHICH == 2) ABC_ASM8("v_mul_hi_u32 v12, v10, v11");
    if (WHICH == 3) ABC_ASM8("v_mul_u32_u24 v12, v10, v11");
    if (WHICH == 4) ABC_ASM8("v_add_u32 v12, v10, v11");
    if (WHICH == 5) ABC_ASM8("v_lshl_add_u64 v[12:13], v[10:11], 0, v[12:13]");
    if (WHICH == 6) ABC_ASM8("v_add_co_u32 v12, vcc, v10, v11");
    if (WHICH == 7) ABC_ASM8("v_cndmask_b32 v12, v10, v11, vcc");
    if (WHICH == 8) ABC_ASM8("v_mad_u32_u24 v12, v10, v11, v12");
    if (WHICH == 9) ABC_ASM8("v_mul_hi_u32_u24 v12, v10, v11");
    if (WHICH == 10) ABC_ASM8("v_fma_f64 v[12:13], v[10:11], v[10:11], v[12:13]");
    if (WHICH == 11) ABC_ASM8("v_mul_f64 v[12:13], v[10:11], v[10:11]");
    if (WHICH == 12) ABC_ASM8("v_add_f64 v[12:13], v[10:11], v[12:13]");
    if (WHICH == 13) ABC_ASM8("v_rndne_f64 v[12:13], v[10:11]");
    if (WHICH == 14) ABC_ASM8("v_cvt_f64_u32 v[12:13], v10");
    if (WHICH == 15) ABC_ASM8("v_and_b32 v12, v10, v11");
    if (WHICH == 16) ABC_ASM8("v_pk_add_f32 v[12:13], v[10:11], v[12:13]");
  }
  u32 r;
  asm volatile("v_mov_b32 %0, v12" : "=v"(r)::"v12");
  sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int microbench_instr(abc_hip_ctx *c, int which, int iters, double *ms) {
  const int blocks = 256 * 4, threads = 256;  // 4 waves per SIMD on every CU
  if (ensure_workspace(c, (size_t)blocks * threads * 8)) return 1;
  u64 *sink = (u64 *)c->ws;
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    ABC_HIP_CHECK(hipEventRecord(c->ev0, c->stream));
    switch (which) {
#define ABC_CASE(W) case W: hipLaunchKernelGGL(k_mb_instr<W>, dim3(blocks), dim3(threads), 0, c->stream, sink, iters); break;
      ABC_CASE(0) ABC_CASE(1) ABC_CASE(2) ABC_CASE(3) ABC_CASE(4) ABC_CASE(5) ABC_CASE(6) ABC_CASE(7) ABC_CASE(8) ABC_CASE(9) ABC_CASE(10) ABC_CASE(11) ABC_CASE(12) ABC_CASE(13) ABC_CASE(14) ABC_CASE(15) ABC_CASE(16)
      default: set_error("microbench: unknown instruction probe"); return 1;
    }
    ABC_HIP_CHECK(hipGetLastError());
    ABC_HIP_CHECK(hipEventRecord(c->ev1, c->stream));
    ABC_HIP_CHECK(hipEventSynchronize(c->ev1));
    float t = 0;
    ABC_HIP_CHECK(hipEventElapsedTime(&t, c->ev0, c->ev1));
    if (t < best) best = t;
  }
  *ms = best;
  return 0;
}
}  // namespace abc

// ---- butterfly issue-rate probes: the compiler's unguarded butterfly vs a hand-scheduled instruction sequence ----
namespace abc {
__global__ __launch_bounds__(256) void k_mb_bfly_cpp(const Mod *mods, u64 *sink, int iters, u64 w, u64 ws) {
  const Mod m = mods[0];
  u64 x[4], y[4];
  for (int k = 0; k < 4; k++) { x[k] = threadIdx.x + 17 * k + 1; y[k] = blockIdx.x + 31 * k + 5; }
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const u64 a = x[k];
      const u64 v = mul_shoup_lazy4(y[k], w, ws, m.q);
      x[k] = a + v;
      y[k] = a + (m.two_q << 1) - v;
    }
  }
  sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = x[0] ^ x[1] ^ x[2] ^ x[3] ^ y[0] ^ y[1] ^ y[2] ^ y[3];
}

// X in v[0+4k:1+4k], Y in v[2+4k:3+4k]; scratch v[16+8k .. 23+8k]; s[20:21]=ws s[22:23]=w s[24:25]=-q s[26:27]=4q+1
#define ABC_BFLY_ASM(X0, X1, Y0, Y1, T0, T1, T2, T3, H0, H1, A0, A1, C0, C1)                                             \
  "v_mul_hi_u32 v" #T0 ", v" #Y1 ", s20\n"                                                                             \
  "v_mul_hi_u32 v" #T2 ", v" #Y0 ", s21\n"                                                                             \
  "v_mad_u64_u32 v[" #H0 ":" #H1 "], s[28:29], v" #Y1 ", s21, v[" #T0 ":" #T1 "]\n"                                    \
  "v_mad_u64_u32 v[" #A0 ":" #A1 "], s[28:29], v" #Y0 ", s22, 0\n"                                                     \
  "v_mad_u64_u32 v[" #C0 ":" #C1 "], s[28:29], v" #Y0 ", s23, 0\n"                                                     \
  "v_lshl_add_u64 v[" #H0 ":" #H1 "], v[" #H0 ":" #H1 "], 0, v[" #T2 ":" #T3 "]\n"                                     \
  "v_mad_u64_u32 v[" #C0 ":" #C1 "], s[28:29], v" #Y1 ", s22, v[" #C0 ":" #C1 "]\n"                                    \
  "v_mad_u64_u32 v[" #A0 ":" #A1 "], s[28:29], v" #H0 ", s24, v[" #A0 ":" #A1 "]\n"                                    \
  "v_mad_u64_u32 v[" #C0 ":" #C1 "], s[28:29], v" #H0 ", s25, v[" #C0 ":" #C1 "]\n"                                    \
  "v_mad_u64_u32 v[" #C0 ":" #C1 "], s[28:29], v" #H1 ", s24, v[" #C0 ":" #C1 "]\n"                                    \
  "v_lshl_add_u64 v[" #Y0 ":" #Y1 "], v[" #X0 ":" #X1 "], 0, s[26:27]\n"                                               \
  "v_add_u32 v" #A1 ", v" #A1 ", v" #C0 "\n"                                                                           \
  "v_lshl_add_u64 v[" #X0 ":" #X1 "], v[" #X0 ":" #X1 "], 0, v[" #A0 ":" #A1 "]\n"                                     \
  "v_not_b32 v" #A0 ", v" #A0 "\n"                                                                                     \
  "v_not_b32 v" #A1 ", v" #A1 "\n"                                                                                     \
  "v_lshl_add_u64 v[" #Y0 ":" #Y1 "], v[" #Y0 ":" #Y1 "], 0, v[" #A0 ":" #A1 "]\n"

__global__ __launch_bounds__(256) void k_mb_bfly_asm(const Mod *mods, u64 *sink, int iters, u64 w, u64 ws) {
  const Mod m = mods[0];
  const u64 nq = 0 - m.q, c4 = (m.two_q << 1) + 1;
  const u32 tid = threadIdx.x, bid = blockIdx.x;
  asm volatile(
      "s_mov_b64 s[20:21], %2\n s_mov_b64 s[22:23], %3\n s_mov_b64 s[24:25], %4\n s_mov_b64 s[26:27], %5\n"
      "v_add_u32 v0, 1, %0\n v_mov_b32 v1, 0\n v_add_u32 v2, 5, %1\n v_mov_b32 v3, 0\n"
      "v_add_u32 v4, 18, %0\n v_mov_b32 v5, 0\n v_add_u32 v6, 36, %1\n v_mov_b32 v7, 0\n"
      "v_add_u32 v8, 35, %0\n v_mov_b32 v9, 0\n v_add_u32 v10, 67, %1\n v_mov_b32 v11, 0\n"
      "v_add_u32 v12, 52, %0\n v_mov_b32 v13, 0\n v_add_u32 v14, 98, %1\n v_mov_b32 v15, 0\n"
      "v_mov_b32 v17, 0\n v_mov_b32 v19, 0\n v_mov_b32 v25, 0\n v_mov_b32 v27, 0\n"
      "v_mov_b32 v33, 0\n v_mov_b32 v35, 0\n v_mov_b32 v41, 0\n v_mov_b32 v43, 0\n" ::"v"(tid),
      "v"(bid), "s"(ws), "s"(w), "s"(nq), "s"(c4)
      : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18",
        "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35",
        "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "s20", "s21", "s22", "s23", "s24",
        "s25", "s26", "s27", "s28", "s29");
  for (int i = 0; i < iters; i++) {
    asm volatile(
        // four butterflies back to back; independent register sets let the hardware overlap their latencies
        ABC_BFLY_ASM(0, 1, 2, 3, 16, 17, 18, 19, 20, 21, 22, 23, 44, 45)
        ABC_BFLY_ASM(4, 5, 6, 7, 24, 25, 26, 27, 28, 29, 30, 31, 46, 47)
        ABC_BFLY_ASM(8, 9, 10, 11, 32, 33, 34, 35, 36, 37, 38, 39, 44, 45)
        ABC_BFLY_ASM(12, 13, 14, 15, 40, 41, 42, 43, 20, 21, 22, 23, 46, 47)
        ::
        : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18",
          "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35",
          "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "s28", "s29");
  }
  u32 r;
  asm volatile("v_xor_b32 %0, v0, v2\n v_xor_b32 %0, %0, v4\n v_xor_b32 %0, %0, v6\n v_xor_b32 %0, %0, v8\n v_xor_b32 %0, %0, v10\n"
               "v_xor_b32 %0, %0, v12\n v_xor_b32 %0, %0, v14\n"
               : "=&v"(r)::"v0", "v2", "v4", "v6", "v8", "v10", "v12", "v14");
  sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// fp64 butterfly for primes below 2^50: v = y*w - rint(y*w/q)*q through two error-free FMA steps (no corrections,
// signed lazy residues), then X = a + v, Y = a - v.  8 DP instructions.
__global__ __launch_bounds__(256) void k_mb_bfly_fp(const Mod *mods, u64 *sink, int iters, double w, double wq) {
  const double q = mods[1].qd;
  double x[4], y[4];
  for (int k = 0; k < 4; k++) { x[k] = threadIdx.x + 17 * k + 1; y[k] = blockIdx.x + 31 * k + 5; }
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const double a = x[k];
      const double v = fp_mul_lazy(y[k], w, wq, q);
      x[k] = a + v;
      y[k] = a - v;
    }
    if ((i & 7) == 7) {
#pragma unroll
      for (int k = 0; k < 4; k++) {  // keep the residues bounded (costs 6/64 extra instructions per butterfly)
        x[k] = __builtin_fma(-__builtin_rint(x[k] * mods[1].qinv), q, x[k]);
        y[k] = __builtin_fma(-__builtin_rint(y[k] * mods[1].qinv), q, y[k]);
      }
    }
  }
  sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = (u64)(long long)(x[0] + x[1] + x[2] + x[3] + y[0] + y[1] + y[2] + y[3]);
}

int microbench_bfly(abc_hip_ctx *c, int which, int iters, double *ms) {
  const int blocks = 256 * 4, threads = 256;
  if (ensure_workspace(c, (size_t)blocks * threads * 8)) return 1;
  u64 *sink = (u64 *)c->ws;
  const u64 q = c->h_mods[0].q, w = q / 3 + 12345, ws = (u64)((((unsigned __int128)w) << 64) / q);
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    ABC_HIP_CHECK(hipEventRecord(c->ev0, c->stream));
    if (which == 0)
      hipLaunchKernelGGL(k_mb_bfly_cpp, dim3(blocks), dim3(threads), 0, c->stream, c->d_mods, sink, iters, w, ws);
    else if (which == 2) {
      const u64 q1 = c->h_mods[1].q, w1 = q1 / 3 + 12345;
      hipLaunchKernelGGL(k_mb_bfly_fp, dim3(blocks), dim3(threads), 0, c->stream, c->d_mods, sink, iters, (double)w1,
                         (double)w1 / (double)q1);
    } else
      hipLaunchKernelGGL(k_mb_bfly_asm, dim3(blocks), dim3(threads), 0, c->stream, c->d_mods, sink, iters, w, ws);
    ABC_HIP_CHECK(hipGetLastError());
    ABC_HIP_CHECK(hipEventRecord(c->ev1, c->stream));
    ABC_HIP_CHECK(hipEventSynchronize(c->ev1));
    float t = 0;
    ABC_HIP_CHECK(hipEventElapsedTime(&t, c->ev0, c->ev1));
    if (t < best) best = t;
  }
  *ms = best;
  return 0;
}
}  // namespace abc

// ---- where does a transform's time go: fp64 2^14-point forward transform with the HBM load and / or store cut out ----
namespace abc {
template <int MODE>  // bit 0: no global loads (synthetic input), bit 1: no global stores (one word per lane instead of 16)
__global__ __launch_bounds__(1024) void k_mb_ntt(DevCtx c, u64 *data, int mid) {
  __shared__ double lds[lds_words(14)];
  const Mod m = c.mods[mid];
  const FpTable t = fp_table(c, mid);
  u64 *base = data + (size_t)blockIdx.x * c.n;
  double sink = 0.0;
  ntt_fwd_block_a<14, FpArith>(
      lds, [&](int, int i) { return (MODE & 1) ? (double)(i ^ (int)blockIdx.x) : fp_from_u64(base[i]); },
      [&](int, int i, double v) {
        if (MODE & 2)
          sink += v;
        else
          base[i] = fp_to_canon(v, m.qd, m.qinv);
      },
      t, m, 0, 0);
  if ((MODE & 2) && sink == 12345.678) base[threadIdx.x] = 1;
}

// staggered start: the first workgroup on each CU waits a pseudo-random fraction of one transform period, so the CUs
// stop marching through their load / compute / store phases in lockstep
__global__ __launch_bounds__(1024) void k_mb_ntt_stagger(DevCtx c, u64 *data, int mid, int ticks) {
  __shared__ double lds[lds_words(14)];
  if (blockIdx.x < 256) {
    const unsigned long long t0 = wall_clock64();
    const unsigned long long d = (unsigned long long)((blockIdx.x * 37u) & 255u) * (unsigned)ticks / 256u;
    while (wall_clock64() - t0 < d) __builtin_amdgcn_s_sleep(8);
  }
  const Mod m = mod_at(c, mid);
  const FpTable t = fp_table(c, mid);
  u64 *base = data + (size_t)blockIdx.x * c.n;
  ntt_fwd_block_a<14, FpArith>(
      lds, [&](int, int i) { return fp_from_u64(base[i]); }, [&](int, int i, double v) { base[i] = fp_to_canon(v, m.qd, m.qinv); },
      t, m, 0, 0);
}

// 4096-point transforms (four workgroups per CU): does a phase offset between co-resident workgroups let one's HBM
// phases hide under another's arithmetic?  ticks = 0: all start together.
__global__ __launch_bounds__(256) void k_mb_ntt12(DevCtx c, u64 *data, int mid, int ticks) {
  __shared__ double lds[lds_words(12)];
  if (ticks && blockIdx.x < 1024) {
    const unsigned long long t0 = wall_clock64();
    const unsigned long long d = (unsigned long long)((blockIdx.x * 37u) & 255u) * (unsigned)ticks / 256u;
    while (wall_clock64() - t0 < d) __builtin_amdgcn_s_sleep(8);
  }
  const Mod m = mod_at(c, mid);
  const FpTable t = fp_table(c, mid);
  u64 *base = data + (size_t)blockIdx.x * 4096;
  ntt_fwd_block_a<12, FpArith>(
      lds, [&](int, int i) { return fp_from_u64(base[i]); }, [&](int, int i, double v) { base[i] = fp_to_canon(v, m.qd, m.qinv); },
      t, m, 2, (int)(blockIdx.x & 3));
}

// persistent form: 256 x PER workgroups walk the transforms; the next transform's 16 words per lane are fetched into
// registers before the current one is computed
template <bool PREFETCH>
__global__ __launch_bounds__(1024) void k_mb_ntt_persistent(DevCtx c, u64 *data, int mid, int total) {
  __shared__ double lds[lds_words(14)];
  const Mod m = c.mods[mid];
  const FpTable t = fp_table(c, mid);
  const int tid = threadIdx.x;
  u64 raw[16];
  int w = blockIdx.x;
  if (PREFETCH && w < total) {
#pragma unroll
    for (int k = 0; k < 16; k++) raw[k] = data[(size_t)w * c.n + (k << 10) + tid];
  }
  for (; w < total; w += gridDim.x) {
    u64 *base = data + (size_t)w * c.n;
    double x[16];
    if (PREFETCH) {
#pragma unroll
      for (int k = 0; k < 16; k++) x[k] = fp_from_u64(raw[k]);
      const int wn = w + gridDim.x;
      if (wn < total) {
#pragma unroll
        for (int k = 0; k < 16; k++) raw[k] = data[(size_t)wn * c.n + (k << 10) + tid];
      }
    }
    block_sync_lds();
    ntt_fwd_block_a<14, FpArith>(
        lds, [&](int r, int i) { return PREFETCH ? x[r] : fp_from_u64(base[i]); },
        [&](int, int i, double v) { base[i] = fp_to_canon(v, m.qd, m.qinv); }, t, m, 0, 0);
  }
}

// split form: two 512-thread workgroups per transform, each folds stage 0 into its load and runs the remaining 13
// stages on its half (68 KiB of LDS, so two workgroups share a CU and their HBM phases overlap the other's compute)
__global__ __launch_bounds__(512) void k_mb_ntt_split(DevCtx c, const u64 *src, u64 *dst, int mid, int pair_stride) {
  __shared__ double lds[lds_words(13)];
  const Mod m = c.mods[mid];
  const FpTable t = fp_table(c, mid);
  // workgroups w and w + pair_stride work on the two halves of one transform
  const int h = (blockIdx.x / pair_stride) & 1;
  const size_t limb = (size_t)(blockIdx.x / (2 * pair_stride)) * pair_stride + (blockIdx.x % pair_stride);
  const u64 *base = src + limb * c.n;
  u64 *out = dst + limb * c.n + ((size_t)h << 13);
  const f64x2 w0 = tw_load(t.tw + 1);
  const double q = m.qd;
  if (h == 0)
    ntt_fwd_block_a<13, FpArith>(
        lds, [&](int, int i) { return fp_from_u64(base[i]) + fp_mul_lazy(fp_from_u64(base[i + 8192]), w0.x, w0.y, q); },
        [&](int, int i, double v) { out[i] = fp_to_canon(v, m.qd, m.qinv); }, t, m, 1, 0);
  else
    ntt_fwd_block_a<13, FpArith>(
        lds, [&](int, int i) { return fp_from_u64(base[i]) - fp_mul_lazy(fp_from_u64(base[i + 8192]), w0.x, w0.y, q); },
        [&](int, int i, double v) { out[i] = fp_to_canon(v, m.qd, m.qinv); }, t, m, 1, 1);
}

// memory-only references: the transform's HBM access pattern with no arithmetic, and a plain streaming copy
__global__ __launch_bounds__(1024) void k_mb_copy_pattern(DevCtx c, const u64 *src, u64 *dst) {
  const u64 *base = src + (size_t)blockIdx.x * c.n;
  u64 *out = dst + (size_t)blockIdx.x * c.n;
  const int tid = threadIdx.x;
  u64 x[16];
#pragma unroll
  for (int k = 0; k < 16; k++) x[k] = base[(k << 10) + tid];
#pragma unroll
  for (int g = 0; g < 4; g++) {
    u64x2 *o = reinterpret_cast<u64x2 *>(out + 4 * (tid + 1024 * g));
    o[0] = u64x2{x[4 * g] + 1, x[4 * g + 1] + 1};
    o[1] = u64x2{x[4 * g + 2] + 1, x[4 * g + 3] + 1};
  }
}
__global__ __launch_bounds__(256) void k_mb_copy_stream(const u64x2 *src, u64x2 *dst, size_t n2) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
    u64x2 v = src[i];
    v.x += 1;
    dst[i] = v;
  }
}

// chip-level overlap check: an arithmetic-only kernel (fp64 FMA chains, 1 workgroup of 256 threads x 4 per CU) and a
// streaming copy on two streams: alone (which = 20, 21) and side by side (22).  iters scales both.
int microbench_overlap(abc_hip_ctx *c, int which, int iters, double *ms) {
  const size_t words = (size_t)64 << 20;  // 512 MiB source + 512 MiB destination
  if (ensure_workspace(c, 2 * words * 8 + (size_t)1024 * 256 * 8)) return 1;
  u64 *src = (u64 *)c->ws, *dst = src + words, *sink = dst + words;
  hipStream_t s0 = c->lane[0], s1 = c->lane[1];
  ABC_HIP_CHECK(hipStreamSynchronize(c->stream));
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    ABC_HIP_CHECK(hipEventRecord(c->ev0, c->stream));
    ABC_HIP_CHECK(hipEventRecord(c->lane_fork, c->stream));
    ABC_HIP_CHECK(hipStreamWaitEvent(s0, c->lane_fork, 0));
    ABC_HIP_CHECK(hipStreamWaitEvent(s1, c->lane_fork, 0));
    if (which != 21)
      hipLaunchKernelGGL(k_mb_fma64, dim3(1024), dim3(256), 0, s0, c->d_mods, sink, iters * 4096);
    if (which != 20)
      for (int r = 0; r < iters; r++)
        hipLaunchKernelGGL(k_mb_copy_stream, dim3(256 * 16), dim3(256), 0, s1, (const u64x2 *)src, (u64x2 *)dst, words / 2);
    ABC_HIP_CHECK(hipGetLastError());
    ABC_HIP_CHECK(hipEventRecord(c->lane_join[0], s0));
    ABC_HIP_CHECK(hipEventRecord(c->lane_join[1], s1));
    ABC_HIP_CHECK(hipStreamWaitEvent(c->stream, c->lane_join[0], 0));
    ABC_HIP_CHECK(hipStreamWaitEvent(c->stream, c->lane_join[1], 0));
    ABC_HIP_CHECK(hipEventRecord(c->ev1, c->stream));
    ABC_HIP_CHECK(hipEventSynchronize(c->ev1));
    float t = 0;
    ABC_HIP_CHECK(hipEventElapsedTime(&t, c->ev0, c->ev1));
    if (t < best) best = t;
  }
  *ms = best;
  return 0;
}

int microbench_ntt(abc_hip_ctx *c, int which, int iters, double *ms) {
  if (which >= 20) return microbench_overlap(c, which, iters, ms);
  if (c->logn != 14) { set_error("microbench: transform probes need N = 2^14"); return 1; }
  const int limbs = iters;  // number of transforms in the launch
  if (ensure_workspace(c, (size_t)2 * limbs * c->n * 8)) return 1;
  u64 *d = (u64 *)c->ws, *d2 = d + (size_t)limbs * c->n;
  ABC_HIP_CHECK(hipMemsetAsync(d, 0, (size_t)limbs * c->n * 8, c->stream));
  float best = 1e30f;
  const int mid = 1;
  for (int rep = 0; rep < 4; rep++) {
    ABC_HIP_CHECK(hipEventRecord(c->ev0, c->stream));
    switch (which) {
      case 0: hipLaunchKernelGGL(k_mb_ntt<0>, dim3(limbs), dim3(1024), 0, c->stream, c->dc, d, mid); break;
      case 1: hipLaunchKernelGGL(k_mb_ntt<1>, dim3(limbs), dim3(1024), 0, c->stream, c->dc, d, mid); break;
      case 2: hipLaunchKernelGGL(k_mb_ntt<2>, dim3(limbs), dim3(1024), 0, c->stream, c->dc, d, mid); break;
      case 3: hipLaunchKernelGGL(k_mb_ntt<3>, dim3(limbs), dim3(1024), 0, c->stream, c->dc, d, mid); break;
      case 4: hipLaunchKernelGGL(k_mb_ntt_persistent<false>, dim3(256), dim3(1024), 0, c->stream, c->dc, d, mid, limbs); break;
      case 6: hipLaunchKernelGGL(k_mb_ntt_split, dim3(2 * limbs), dim3(512), 0, c->stream, c->dc, d, d2, mid, 1); break;
      case 7: hipLaunchKernelGGL(k_mb_ntt_split, dim3(2 * limbs), dim3(512), 0, c->stream, c->dc, d, d2, mid, 8); break;
      case 8: hipLaunchKernelGGL(k_mb_copy_pattern, dim3(limbs), dim3(1024), 0, c->stream, c->dc, d, d2); break;
      case 9: hipLaunchKernelGGL(k_mb_copy_stream, dim3(256 * 16), dim3(256), 0, c->stream, (const u64x2 *)d, (u64x2 *)d2,
                                 (size_t)limbs * c->n / 2); break;
      case 10: hipLaunchKernelGGL(k_mb_ntt_stagger, dim3(limbs), dim3(1024), 0, c->stream, c->dc, d, mid, 600); break;
      case 11: hipLaunchKernelGGL(k_mb_ntt_stagger, dim3(limbs), dim3(1024), 0, c->stream, c->dc, d, mid, 1200); break;
      case 12: hipLaunchKernelGGL(k_mb_ntt_stagger, dim3(limbs), dim3(1024), 0, c->stream, c->dc, d, mid, 1700); break;
      case 13: hipLaunchKernelGGL(k_mb_ntt12, dim3(limbs * 4), dim3(256), 0, c->stream, c->dc, d, mid, 0); break;
      case 14: hipLaunchKernelGGL(k_mb_ntt12, dim3(limbs * 4), dim3(256), 0, c->stream, c->dc, d, mid, 200); break;
      case 15: hipLaunchKernelGGL(k_mb_ntt12, dim3(limbs * 4), dim3(256), 0, c->stream, c->dc, d, mid, 430); break;
      case 16: hipLaunchKernelGGL(k_mb_ntt12, dim3(limbs * 4), dim3(256), 0, c->stream, c->dc, d, mid, 900); break;
      case 5: hipLaunchKernelGGL(k_mb_ntt_persistent<true>, dim3(256), dim3(1024), 0, c->stream, c->dc, d, mid, limbs); break;
      default: set_error("microbench: unknown transform probe"); return 1;
    }
    ABC_HIP_CHECK(hipGetLastError());
    ABC_HIP_CHECK(hipEventRecord(c->ev1, c->stream));
    ABC_HIP_CHECK(hipEventSynchronize(c->ev1));
    float t = 0;
    ABC_HIP_CHECK(hipEventElapsedTime(&t, c->ev0, c->ev1));
    if (rep && t < best) best = t;
  }
  *ms = best;
  return 0;
}
}  // namespace abc
#else
namespace abc {
int microbench(abc_hip_ctx *, int, int, double *) {
  set_error("this libabc_hip.so was built without the probe kernels (python -m abc_amd.build --microbench)");
  return 1;
}
}  // namespace abc
#endif
