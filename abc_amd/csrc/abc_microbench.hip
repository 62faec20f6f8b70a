// abc_microbench.hip -- issue-rate probes for the two arithmetic pipes the NTT can run on:
// the 32-bit integer multiplier (v_mad_u64_u32 / v_mul_hi_u32) and the fp64 FMA pipe.  They decide
// which modular-multiplication scheme the hot kernels use (DESIGN.md "Arithmetic roofline").
#include "abc_context.hpp"

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
int microbench(abc_hip_ctx *c, int which, int iters, double *ms) {
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
      ABC_CASE(0) ABC_CASE(1) ABC_CASE(2) ABC_CASE(3) ABC_CASE(4) ABC_CASE(5) ABC_CASE(6) ABC_CASE(7) ABC_CASE(8) ABC_CASE(9)
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
