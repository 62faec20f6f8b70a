// abc_ntt.hpp -- LDS-staged negacyclic NTT / INTT building blocks for gfx950.
//
// One workgroup transforms one contiguous block of 2^LB coefficients (LB = 10..14: 8..128 KiB of the
// CU's 160 KiB LDS) of one RNS limb.  Each thread keeps 16 coefficients in VGPRs and performs up to
// four butterfly stages (radix-16) between two LDS exchanges, so a 2^14-point transform touches LDS
// three times and HBM exactly once in and once out.  Ordering follows the reference's SEAL
// dependency: forward = Cooley-Tukey, natural in -> bit-reversed out; inverse = Gentleman-Sande,
// bit-reversed in -> natural out, with twiddles tw[m+i] = psi^bitrev(m+i) (psi = minimal primitive
// 2N-th root) -- the layout every NTT-form ciphertext, key and plaintext of the reference's SEAL
// runtime uses (call sites: src/runtime/SealCiphertext.cpp:104-105,122-123,159,196).
// Butterflies are Harvey lazy ([0,4q) forward, [0,2q) inverse); callers canonicalise in the store
// functor, so outputs are bit-exact with the oracle.
//
// A block may be a sub-transform of a larger N (N = 2^15, 2^16 do not fit LDS): the caller passes the
// number of stages already done (S0) and the block index b, and the first S0 stages are done by the
// strided global pass in abc_ntt.hip.
#pragma once

#include "abc_modarith.hpp"

namespace abc {

// LDS padding: 4 extra words every 64 keeps the stride-2^k accesses of the middle passes
// conflict-free for ds_read_b64/ds_write_b64 (bank = (addr/4) mod 64).
__device__ __forceinline__ int lds_pad(int i) { return i + ((i >> 6) << 2); }
constexpr int lds_words(int lb) { return (1 << lb) + ((1 << lb) >> 6) * 4; }

// Twiddle tables are interleaved {w, floor(w * 2^64 / q)} pairs so that one 16-byte load (s_load_dwordx4 where the
// index is wave-uniform, global_load_dwordx4 otherwise) fetches everything a butterfly needs.
struct NttTable {
  const u64x2 *tw;   // [N] forward twiddles {w, Shoup quotient}, bit-reversed order: tw[m+i].x = psi^bitrev(m+i)
  const u64x2 *itw;  // [N] inverse twiddles (itw[m+i].x = tw[m+i].x^-1)
};

// ---- one radix-2^R register pass over NG = 16>>R groups -------------------------------------------
// Forward (CT): stages S..S+R-1 of the block-local transform.
// GUARD = false: no per-stage correction of X and the cheaper quotient estimate (mul_shoup_lazy4, product in
// [0,4q)); every stage then adds at most 4q to a value, so after all logN <= 14 stages of a block a canonical
// input stays below (4*14 + 1) q = 57q -- usable whenever that fits 64 bits (q <= 57 bits), which holds for every
// SEAL default prime up to N = 32768 and for the CKKS chains here; the 61-bit BEHZ primes (and 58..60-bit user
// primes) take the guarded form.
template <int LB, int S, int R, bool UNIFORM, bool GUARD = true>
__device__ __forceinline__ void fwd_pass(u64 (&x)[16], const int (&hi)[16 >> R], const NttTable &t, u64 q, u64 two_q,
                                         int S0, int b) {
  constexpr int NG = 16 >> R;
#pragma unroll
  for (int u = 0; u < R; u++) {
    const int half = 1 << (R - 1 - u);
#pragma unroll
    for (int g = 0; g < NG; g++) {
      // twiddle index of butterfly block: global stage S0+S+u
      int base = (1 << (S0 + S + u)) + (b << (S + u)) + (hi[g] << u);
#pragma unroll
      for (int k = 0; k < (1 << R); k++) {
        if (k & half) continue;
        int idx = base + (k >> (R - u));
        const u64x2 tp = t.tw[idx];
        const u64 w = tp.x, ws = tp.y;
        u64 &X = x[g * (1 << R) + k];
        u64 &Y = x[g * (1 << R) + (k | half)];
        if (GUARD) {
          const u64 a = csub(X, two_q);
          const u64 v = mul_shoup_lazy(Y, w, ws, q);
          X = a + v;
          Y = a + two_q - v;
        } else {
          const u64 a = X;
          const u64 v = mul_shoup_lazy4(Y, w, ws, q);
          X = a + v;
          Y = a + (two_q << 1) - v;
        }
      }
    }
  }
}

// Inverse (GS): stages S+R-1 .. S (reverse order). Inputs/outputs in [0,2q).
template <int LB, int S, int R, bool UNIFORM>
__device__ __forceinline__ void inv_pass(u64 (&x)[16], const int (&hi)[16 >> R], const NttTable &t, u64 q, u64 two_q,
                                         int S0, int b) {
  constexpr int NG = 16 >> R;
#pragma unroll
  for (int u = R - 1; u >= 0; u--) {
    const int half = 1 << (R - 1 - u);
#pragma unroll
    for (int g = 0; g < NG; g++) {
      int base = (1 << (S0 + S + u)) + (b << (S + u)) + (hi[g] << u);
#pragma unroll
      for (int k = 0; k < (1 << R); k++) {
        if (k & half) continue;
        int idx = base + (k >> (R - u));
        const u64x2 tp = t.itw[idx];
        const u64 w = tp.x, ws = tp.y;
        u64 &X = x[g * (1 << R) + k];
        u64 &Y = x[g * (1 << R) + (k | half)];
        u64 a = X, c = Y;
        X = csub(a + c, two_q);
        Y = mul_shoup_lazy(a + two_q - c, w, ws, q);
      }
    }
  }
}

// Index bookkeeping of one pass: group p has lo = p mod G, hi = p / G (G = 2^(LB-S-R)); its k-th element sits at
// (hi << (LB-S)) + (k << logG) + lo.  Group ownership is wave-contiguous: lane l of wave w owns groups
// w*64*NG + g*64 + l.  For the first pass (NG = 1) that is simply p = tid; for every later pass (S >= LB-10) a
// wave's 64*NG groups cover exactly the contiguous block of 1024 coefficients [1024 w, 1024 (w+1)), so those
// passes only touch LDS words written by the same wave and need no workgroup barrier (see wave_sync).
template <int LB, int S, int R>
struct PassIdx {
  static constexpr int T = (1 << LB) / 16;
  static constexpr int NG = 16 >> R;
  static constexpr int LOGG = LB - S - R;
  static constexpr int G = 1 << LOGG;
  static constexpr bool UNIFORM = (G >= 64);  // all 64 lanes of a wave share hi
  __device__ __forceinline__ static void groups(int tid, int (&hi)[NG], int (&lo)[NG]) {
    // the wave index as a scalar: where hi is wave-uniform (G >= 64) every twiddle index then stays in SGPRs
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
#pragma unroll
    for (int g = 0; g < NG; g++) {
      if (UNIFORM) {
        const int pw = wave * (64 * NG) + g * 64;  // scalar part of p; the lane part cannot reach bit LOGG
        hi[g] = pw >> LOGG;
        lo[g] = (pw + lane) & (G - 1);
      } else {
        const int p = wave * (64 * NG) + g * 64 + lane;
        hi[g] = p >> LOGG;
        lo[g] = p & (G - 1);
      }
    }
  }
  __device__ __forceinline__ static int elem(int hi, int lo, int k) { return (hi << (LB - S)) + (k << LOGG) + lo; }
};

template <int LB, int S, int R>
__device__ __forceinline__ void lds_load(const u64 *lds, u64 (&x)[16], const int (&hi)[16 >> R], const int (&lo)[16 >> R]) {
  using P = PassIdx<LB, S, R>;
#pragma unroll
  for (int g = 0; g < P::NG; g++)
#pragma unroll
    for (int k = 0; k < (1 << R); k++) x[g * (1 << R) + k] = lds[lds_pad(P::elem(hi[g], lo[g], k))];
}
template <int LB, int S, int R>
__device__ __forceinline__ void lds_store(u64 *lds, const u64 (&x)[16], const int (&hi)[16 >> R], const int (&lo)[16 >> R]) {
  using P = PassIdx<LB, S, R>;
#pragma unroll
  for (int g = 0; g < P::NG; g++)
#pragma unroll
    for (int k = 0; k < (1 << R); k++) lds[lds_pad(P::elem(hi[g], lo[g], k))] = x[g * (1 << R) + k];
}

// Exchange through LDS between two passes that are local to a wave: DS operations of one wave execute in
// issue order, so only the compiler must be kept from reordering them across this point.
// Both syncs fence the LDS ("local") address space only, so the compiler may move the global twiddle loads of the
// next pass above them and their latency hides under the current pass.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}
__device__ __forceinline__ void block_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Pass schedules (sum = LB, last forward pass has R = 2 so each lane ends with 32-byte contiguous runs).
template <int LB> struct Sched;
template <> struct Sched<10> { static constexpr int R0 = 4, R1 = 4, R2 = 2, R3 = 0; };
template <> struct Sched<11> { static constexpr int R0 = 4, R1 = 3, R2 = 2, R3 = 2; };
template <> struct Sched<12> { static constexpr int R0 = 4, R1 = 4, R2 = 2, R3 = 2; };
template <> struct Sched<13> { static constexpr int R0 = 4, R1 = 4, R2 = 3, R3 = 2; };
template <> struct Sched<14> { static constexpr int R0 = 4, R1 = 4, R2 = 4, R3 = 2; };

// ---- forward block transform -----------------------------------------------------------------------
// load(r, i)  -> u64 in [0,4q)  coefficient i (block-local natural index) for register slot r
// store(r, i, v)               v in [0,4q) lazily reduced value of output slot i (block-local,
//                              bit-reversed order) held in register slot r (r is a compile-time
//                              constant after unrolling, so callers may index register arrays with it)
// The final register layout is PassIdx<LB, LB-2, 2>: slot r = 4g+k holds element 4*(tid + T*g) + k.
template <int LB, bool GUARD = true, class Load, class Store>
__device__ __forceinline__ void ntt_fwd_block(u64 *lds, Load load, Store store, const NttTable &t, const Mod &m, int S0,
                                              int b) {
  using SC = Sched<LB>;
  const int tid = threadIdx.x;
  const u64 q = m.q, two_q = m.two_q;
  u64 x[16];
  {  // pass 0: global -> regs -> LDS
    constexpr int S = 0, R = SC::R0;
    using P = PassIdx<LB, S, R>;
    int hi[P::NG], lo[P::NG];
    P::groups(tid, hi, lo);
#pragma unroll
    for (int g = 0; g < P::NG; g++)
#pragma unroll
      for (int k = 0; k < (1 << R); k++) x[g * (1 << R) + k] = load(g * (1 << R) + k, P::elem(hi[g], lo[g], k));
    fwd_pass<LB, S, R, P::UNIFORM, GUARD>(x, hi, t, q, two_q, S0, b);
    lds_store<LB, S, R>(lds, x, hi, lo);
  }
  block_sync_lds();
  {  // pass 1
    constexpr int S = SC::R0, R = SC::R1;
    using P = PassIdx<LB, S, R>;
    int hi[P::NG], lo[P::NG];
    P::groups(tid, hi, lo);
    lds_load<LB, S, R>(lds, x, hi, lo);
    fwd_pass<LB, S, R, P::UNIFORM, GUARD>(x, hi, t, q, two_q, S0, b);
    if constexpr (SC::R3 == 0 && SC::R2 == 0) {
#pragma unroll
      for (int g = 0; g < P::NG; g++)
#pragma unroll
        for (int k = 0; k < (1 << R); k++) store(g * (1 << R) + k, P::elem(hi[g], lo[g], k), x[g * (1 << R) + k]);
      return;
    } else {
      lds_store<LB, S, R>(lds, x, hi, lo);  // in place: a thread rewrites exactly the words it read
    }
  }
  wave_sync();  // passes after the first are local to a wave (PassIdx): no workgroup barrier
  {  // pass 2
    constexpr int S = SC::R0 + SC::R1, R = SC::R2;
    using P = PassIdx<LB, S, R>;
    int hi[P::NG], lo[P::NG];
    P::groups(tid, hi, lo);
    lds_load<LB, S, R>(lds, x, hi, lo);
    fwd_pass<LB, S, R, P::UNIFORM, GUARD>(x, hi, t, q, two_q, S0, b);
    if constexpr (SC::R3 == 0) {
#pragma unroll
      for (int g = 0; g < P::NG; g++)
#pragma unroll
        for (int k = 0; k < (1 << R); k++) store(g * (1 << R) + k, P::elem(hi[g], lo[g], k), x[g * (1 << R) + k]);
      return;
    } else {
      lds_store<LB, S, R>(lds, x, hi, lo);  // in place: a thread rewrites exactly the words it read
    }
  }
  if constexpr (SC::R3 != 0) {
    wave_sync();
    constexpr int S = SC::R0 + SC::R1 + SC::R2, R = SC::R3;
    using P = PassIdx<LB, S, R>;
    int hi[P::NG], lo[P::NG];
    P::groups(tid, hi, lo);
    lds_load<LB, S, R>(lds, x, hi, lo);
    fwd_pass<LB, S, R, P::UNIFORM, GUARD>(x, hi, t, q, two_q, S0, b);
#pragma unroll
    for (int g = 0; g < P::NG; g++)
#pragma unroll
      for (int k = 0; k < (1 << R); k++) store(g * (1 << R) + k, P::elem(hi[g], lo[g], k), x[g * (1 << R) + k]);
  }
}

// ---- inverse block transform -----------------------------------------------------------------------
// load(r, i)  -> u64 in [0,2q)  input slot i (block-local, bit-reversed order); same register layout
//                              as the forward transform's final one
// store(r, i, v)               v in [0,2q): coefficient i BEFORE the N^-1 scaling (caller scales:
//                              for a sub-block the scaling belongs to the final strided pass)
template <int LB, class Load, class Store>
__device__ __forceinline__ void ntt_inv_block(u64 *lds, Load load, Store store, const NttTable &t, const Mod &m, int S0,
                                              int b) {
  using SC = Sched<LB>;
  const int tid = threadIdx.x;
  const u64 q = m.q, two_q = m.two_q;
  u64 x[16];
  constexpr int SA = SC::R0, SB = SC::R0 + SC::R1, SCc = SC::R0 + SC::R1 + SC::R2;
  if constexpr (SC::R3 != 0) {
    constexpr int S = SCc, R = SC::R3;
    using P = PassIdx<LB, S, R>;
    int hi[P::NG], lo[P::NG];
    P::groups(tid, hi, lo);
#pragma unroll
    for (int g = 0; g < P::NG; g++)
#pragma unroll
      for (int k = 0; k < (1 << R); k++) x[g * (1 << R) + k] = load(g * (1 << R) + k, P::elem(hi[g], lo[g], k));
    inv_pass<LB, S, R, P::UNIFORM>(x, hi, t, q, two_q, S0, b);
    lds_store<LB, S, R>(lds, x, hi, lo);
    wave_sync();  // all passes but the last are local to a wave
  }
  {
    constexpr int S = SB, R = SC::R2;
    using P = PassIdx<LB, S, R>;
    int hi[P::NG], lo[P::NG];
    P::groups(tid, hi, lo);
    if constexpr (SC::R3 != 0) {
      lds_load<LB, S, R>(lds, x, hi, lo);
    } else {
#pragma unroll
      for (int g = 0; g < P::NG; g++)
#pragma unroll
        for (int k = 0; k < (1 << R); k++) x[g * (1 << R) + k] = load(g * (1 << R) + k, P::elem(hi[g], lo[g], k));
    }
    inv_pass<LB, S, R, P::UNIFORM>(x, hi, t, q, two_q, S0, b);
    lds_store<LB, S, R>(lds, x, hi, lo);
    wave_sync();
  }
  {
    constexpr int S = SA, R = SC::R1;
    using P = PassIdx<LB, S, R>;
    int hi[P::NG], lo[P::NG];
    P::groups(tid, hi, lo);
    lds_load<LB, S, R>(lds, x, hi, lo);
    inv_pass<LB, S, R, P::UNIFORM>(x, hi, t, q, two_q, S0, b);
    lds_store<LB, S, R>(lds, x, hi, lo);
    block_sync_lds();
  }
  {
    constexpr int S = 0, R = SC::R0;
    using P = PassIdx<LB, S, R>;
    int hi[P::NG], lo[P::NG];
    P::groups(tid, hi, lo);
    lds_load<LB, S, R>(lds, x, hi, lo);
    inv_pass<LB, S, R, P::UNIFORM>(x, hi, t, q, two_q, S0, b);
#pragma unroll
    for (int g = 0; g < P::NG; g++)
#pragma unroll
      for (int k = 0; k < (1 << R); k++) store(g * (1 << R) + k, P::elem(hi[g], lo[g], k), x[g * (1 << R) + k]);
  }
}

// lazily reduced [0,4q) -> [0,q)
__device__ __forceinline__ u64 canon4(u64 v, const Mod &m) { return csub(csub(v, m.two_q), m.q); }
// output of an unguarded forward transform, [0,64q) -> [0,q)
__device__ __forceinline__ u64 canon32(u64 v, const Mod &m) {
  v = csub(v, m.two_q << 4);
  v = csub(v, m.two_q << 3);
  v = csub(v, m.two_q << 2);
  v = csub(v, m.two_q << 1);
  return canon4(v, m);
}
template <bool GUARD>
__device__ __forceinline__ u64 canon_fwd(u64 v, const Mod &m) { return GUARD ? canon4(v, m) : canon32(v, m); }
// a modulus may take the unguarded butterflies when 64 q <= 2^64
__host__ __device__ __forceinline__ bool unguarded_ok(u32 bits) { return bits <= 57; }
// scale by N^-1 and canonicalise (input [0,2q) or any 64-bit value)
__device__ __forceinline__ u64 scale_inv_n(u64 v, const Mod &m) { return mul_shoup(v, m.inv_n, m.inv_n_s, m.q); }

}  // namespace abc
