// abc_context.hpp -- host-side context of libabc_hip.so and the structs shared with device code.
#pragma once

#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <unordered_map>
#include <string>
#include <vector>

#include "abc_modarith.hpp"
#include "abc_ntt.hpp"

namespace abc {

constexpr int kMaxLimbs = 16;

// limb index -> modulus id (index into DevCtx::mods / twiddle tables)
struct LimbMap {
  int id[kMaxLimbs + 1];
};

// small read-only constants, one instance per context in device memory
struct DevConst {
  // key switching / modulus switching
  u64 inv_special[kMaxLimbs], inv_special_s[kMaxLimbs];        // q_special^-1 mod q_j (+Shoup)
  u64 inv_qlast[kMaxLimbs][kMaxLimbs], inv_qlast_s[kMaxLimbs][kMaxLimbs];  // [l][j] = q_l^-1 mod q_j
  u64 special_mod_q[kMaxLimbs], special_mod_q_s[kMaxLimbs];
  double inv_special_c[kMaxLimbs], inv_special_cq[kMaxLimbs];  // fp64 path: centred value and value / q_j
  double special_c[kMaxLimbs], special_cq[kMaxLimbs];          // q_special mod q_j, centred, and that / q_j
  double inv_qlast_c[kMaxLimbs][kMaxLimbs], inv_qlast_cq[kMaxLimbs][kMaxLimbs];  // same for inv_qlast (rescale)                                // q_special mod q_j (key generation)
  // BFV plaintext scaling (Evaluator::add_plain / Encryptor)
  u64 q_mod_t, upper_half_threshold, t;
  u64 coeff_div_plain[kMaxLimbs], upper_half_increment[kMaxLimbs];
  // BEHZ
  int nq, nB, nBsk, pad_;
  u64 mtilde_mod_q[kMaxLimbs];
  u64 inv_punct_q[kMaxLimbs];
  u64 q_to_bsk[kMaxLimbs][kMaxLimbs];  // [j][i] (q/q_i) mod Bsk_j
  u64 q_to_mtilde[kMaxLimbs];          // (q/q_i) mod 2^32
  u64 neg_inv_q_mod_mtilde;
  u64 q_mod_bsk[kMaxLimbs], inv_mtilde_mod_bsk[kMaxLimbs], inv_q_mod_bsk[kMaxLimbs];
  u64 inv_punct_B[kMaxLimbs];
  u64 B_to_q[kMaxLimbs][kMaxLimbs];  // [i][b] (B/B_b) mod q_i
  u64 B_to_msk[kMaxLimbs];
  u64 inv_B_mod_msk, B_mod_q[kMaxLimbs];
  u64 t_mod_q[kMaxLimbs], t_mod_bsk[kMaxLimbs];
  // fused constants with Shoup quotients (a constant operand costs half a Barrett multiply):
  u64 ext_q[kMaxLimbs], ext_q_s[kMaxLimbs];          // m~ * (q/q_i)^-1 mod q_i          (BEHZ extend)
  u64 flr_q[kMaxLimbs], flr_q_s[kMaxLimbs];          // t  * (q/q_i)^-1 mod q_i          (BEHZ floor)
  u64 tinvq_bsk[kMaxLimbs], tinvq_bsk_s[kMaxLimbs];  // t * q^-1 mod Bsk_j
  u64 inv_q_mod_bsk_s[kMaxLimbs], inv_mtilde_mod_bsk_s[kMaxLimbs], inv_punct_B_s[kMaxLimbs];
  u64 inv_B_mod_msk_s, B_mod_q_s[kMaxLimbs];
  u64 dec_q[kMaxLimbs], dec_q_s[kMaxLimbs];          // t*gamma * (q/q_i)^-1 mod q_i     (decrypt)
  // Shoup quotients of the base-conversion matrices (lazy dot products: one mulhi per term, no 128-bit sums)
  u64 q_to_bsk_s[kMaxLimbs][kMaxLimbs], B_to_q_s[kMaxLimbs][kMaxLimbs], B_to_msk_s[kMaxLimbs], q_mod_bsk_s[kMaxLimbs];
  // BFV decryption (decrypt_scale_and_round)
  u64 tgamma_mod_q[kMaxLimbs], q_to_t[kMaxLimbs], q_to_gamma[kMaxLimbs];
  u64 neg_inv_q_mod_t, neg_inv_q_mod_gamma, inv_gamma_mod_t, gamma;
};

// fp64 twins of the BEHZ constants ({value centred into (-p/2, p/2], value / p}: the operand pair of fp_mul_lazy), filled
// when every ciphertext prime and every auxiliary prime is below 2^50 (abc_kernels_bfv.hip, k_behz_extend_fp / k_behz_floor_fp)
struct DevConstFp {
  double ext_q[kMaxLimbs][2], flr_q[kMaxLimbs][2], B_mod_q[kMaxLimbs][2];
  double q_to_bsk[kMaxLimbs][kMaxLimbs][2];  // [j][i]
  double B_to_q[kMaxLimbs][kMaxLimbs][2];    // [i][b]
  double q_mod_bsk[kMaxLimbs][2], inv_mtilde_mod_bsk[kMaxLimbs][2], tinvq_bsk[kMaxLimbs][2], inv_q_mod_bsk[kMaxLimbs][2];
  double inv_punct_B[kMaxLimbs][2], B_to_msk[kMaxLimbs][2], inv_B_mod_msk[2];
};

// passed BY VALUE to every kernel
struct DevCtx {
  const Mod *mods;      // [nmods]
  const u64 *tw;        // [nmods][2][N][2]: forward {w, Shoup} pairs, then inverse pairs
  const double *ftw;    // same shape, fp64 twin {w centred, w / q}; filled for primes < 2^50 only
  const DevConst *cst;  //
  const DevConstFp *cstf;  // null unless the BEHZ base is fp64-capable
  const u32 *slot_map;  // [N] BatchEncoder index map (BFV)
  int logn, n;
  int ps;               // scratch limb stride of the split kernels in words: n + pad (HBM channel spread)
  int K, L;             // key-level primes, data limbs
  int id_bsk, id_t, id_gamma, id_mtilde;  // modulus ids: key primes are 0..K-1
};

__device__ __forceinline__ NttTable ntt_table(const DevCtx &c, int mid) {
  const u64 *b = c.tw + (size_t)mid * 4 * c.n;
  NttTable t;
  t.tw = reinterpret_cast<const u64x2 *>(b);
  t.itw = reinterpret_cast<const u64x2 *>(b + 2 * (size_t)c.n);
  return t;
}

// modulus constants through the constant address space (scalar loads wherever the id is wave-uniform; see FpTable)
__device__ __forceinline__ Mod mod_at(const DevCtx &c, int id) {
  const ABC_CONST_AS Mod *p = (const ABC_CONST_AS Mod *)(c.mods + id);
  Mod m;
  m.q = p->q; m.mu = p->mu; m.two_q = p->two_q; m.shift = p->shift; m.bits = p->bits;
  m.inv_n = p->inv_n; m.inv_n_s = p->inv_n_s; m.qd = p->qd; m.qinv = p->qinv;
  m.inv_n_c = p->inv_n_c; m.inv_n_cq = p->inv_n_cq;
  return m;
}
__device__ __forceinline__ FpTable fp_table(const DevCtx &c, int mid) {
  const double *b = c.ftw + (size_t)mid * 4 * c.n;
  FpTable t;
  t.tw = (const ABC_CONST_AS f64x2 *)(b);
  t.itw = (const ABC_CONST_AS f64x2 *)(b + 2 * (size_t)c.n);
  return t;
}

}  // namespace abc

// The opaque C-ABI handle.
struct abc_hip_ctx {
  int scheme = 0, logn = 0, n = 0, K = 0, L = 0, device = 0;
  std::vector<uint64_t> primes;  // key-level chain: data limbs + special
  uint64_t t = 0;
  hipStream_t stream = nullptr;      // stream every operation is enqueued on
  hipStream_t own_stream = nullptr;  // the context's private stream (default)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // two internal lanes let an HBM-streaming kernel of one chunk overlap an ALU-bound transform of another
  static constexpr int kMaxLanes = 4;
  hipStream_t lane[kMaxLanes] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t lane_fork = nullptr, lane_join[kMaxLanes] = {nullptr, nullptr, nullptr, nullptr};
  // moduli: ids 0..K-1 key primes, then Bsk (B_0..B_{nB-1}, m_sk), gamma, t, m_tilde(arith only)
  std::vector<abc::Mod> h_mods;
  std::vector<uint64_t> mod_values;
  int nB = 0, nBsk = 0;
  abc::DevCtx dc{};
  abc::DevConst h_cst{};
  abc::Mod *d_mods = nullptr;
  uint64_t *d_tw = nullptr;
  double *d_ftw = nullptr;
  // caching allocator behind abc_hip_malloc / abc_hip_free (abc_context.hip)
  bool cache_alloc = false;
  std::mutex alloc_mu;  // guards the three members below
  size_t cached_bytes = 0, cache_cap = (size_t)8 << 30;
  std::unordered_map<size_t, std::vector<void *>> free_blocks;  // size -> cached blocks
  std::unordered_map<void *, size_t> block_size;                 // every live block handed out by abc_hip_malloc
  // Blocks a recorded circuit (abc_hip_graph_*) may have baked into its kernel arguments are PINNED to it: while the graph exists
  // they never go back to the free list, whoever frees them.  Pinned = every block this context had handed out and not yet got
  // back when abc_hip_graph_end ran (a superset of what the sequence read: operands that existed before the capture -- cached
  // plaintexts, inputs -- are covered without tracing every pointer argument) plus every block the capture itself took from the
  // cache.  pin: block -> graphs that own it (kCapturing stands for the capture in progress); parked: pinned blocks the caller
  // has already freed (released to the cache when their last graph is destroyed); cap_free: blocks allocated AND freed during the
  // running capture, reusable inside it (stream order inside the graph).
  bool capture_active = false;
  std::unordered_map<void *, std::vector<void *>> pin;
  std::unordered_map<void *, bool> parked;
  std::unordered_map<size_t, std::vector<void *>> cap_free;
  std::unordered_map<void *, bool> cap_born;  // allocated during the running capture
  bool behz_fp = false;  // BFV: 50-bit BEHZ auxiliary base and fp64 base-conversion kernels
  bool use_fp = true;  // fp64 transforms for primes < 2^50 (ABC_HIP_NO_FP64=1 forces the integer path)
  // Path switches (A/B timing and the parity tests of every fallback): the ABC_HIP_* environment variables are read
  // ONCE, when the context is created (abc_hip_ctx_reload_env re-reads them), never on the per-operation path.
  struct Switches {
    bool no_fused = false, no_split = false, no_split4 = false, no_isplit = false, no_gsplit = false, no_lean_front = false, no_bsplit = false, no_mixed = false, no_pack = false, no_key_twin = false, no_special8x2 = false, no_bmul = false, no_bmul_mid = false, no_finish_lds = false, no_iks = false, no_bmul_r6 = false, no_tensor_intt = false;
    bool no_galois_fusion = false;
    size_t chunk = 0, few_limbs = 48, lean_limit = 96, bfv_scratch_mb = 0, pass0_target_limit = 128;
    int lanes = 2;
    unsigned lane_offset_us = 0;
  } sw;
  abc::DevConst *d_cst = nullptr;
  abc::DevConstFp *d_cstf = nullptr;
  uint32_t *d_slot_map = nullptr;
  // keys (device)
  uint64_t *d_sk = nullptr, *d_pk = nullptr, *d_relin = nullptr;
  std::map<uint32_t, uint64_t *> d_galois;
  // fp64 twins of key-switching keys (centred doubles, same layout), built on first use by the fp64 split kernels, dropped when the
  // key they mirror is rewritten (abc_kernels_fused.hip, key_twin)
  std::unordered_map<const uint64_t *, double *> key_twins;
  std::unordered_map<const uint64_t *, uint64_t *> key_shoups;  // Shoup quotients of a key (abc_kernels_eval.hip, key_shoup)
  std::vector<uint32_t> galois_order;
  // workspace (kernel-sequence scratch) and three caller-level arenas (products, rotation ping-pong buffers);
  // all grow on demand and are reused, so steady-state calls perform no hipMalloc / hipFree
  void *ws = nullptr;
  size_t ws_bytes = 0;
  void *aux[3] = {nullptr, nullptr, nullptr};
  size_t aux_bytes[3] = {0, 0, 0};
  size_t limb_words() const { return (size_t)n; }
  size_t key_words() const { return (size_t)L * 2 * K * n; }
};

namespace abc {

void set_error(const std::string &msg);
#define ABC_HIP_CHECK(expr)                                                                         \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess) {                                                                         \
      abc::set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                            \
      return 1;                                                                                     \
    }                                                                                               \
  } while (0)

// workspace: grows on demand (never inside a timed region after warm-up)
int ensure_workspace(abc_hip_ctx *c, size_t bytes);
int ensure_aux(abc_hip_ctx *c, int which, size_t bytes);
void read_switches(abc_hip_ctx *c);

// ---- launchers implemented in the kernel translation units ----
LimbMap key_limb_map(const abc_hip_ctx *c, int nl);  // 0..nl-1 -> data primes, nl -> special prime
// in-place forward / inverse NTT over `limbs` consecutive limbs laid out [groups][nl][N]; limb j of each
// group uses modulus map.id[j % nl]
int launch_ntt_fwd(abc_hip_ctx *c, u64 *d, const LimbMap &map, int nl, size_t total_limbs);
int launch_ks_expand_ntt_fp(abc_hip_ctx *c, const u64 *tcoef, size_t tstride, u64 *dec, const LimbMap &map, int nl, size_t count);
int launch_ntt_fwd_from2(abc_hip_ctx *c, const u64 *src, const u64 *src2, u64 *d, const LimbMap &map, int nl, size_t total_limbs);
int launch_ntt_fwd_from(abc_hip_ctx *c, const u64 *src, u64 *d, const LimbMap &map, int nl, size_t total_limbs);  // out of place
int launch_ntt_inv(abc_hip_ctx *c, u64 *d, const LimbMap &map, int nl, size_t total_limbs);
int launch_ntt_inv_strided_part(abc_hip_ctx *c, u64 *d, const LimbMap &map, int nl, size_t total_limbs, bool integer_only = false);
int big_block_log(void);

int launch_addsub(abc_hip_ctx *c, const u64 *a, const u64 *b, u64 *out, int nl, size_t polys, int op);  // 0 add 1 sub 2 neg
int launch_ckks_tensor(abc_hip_ctx *c, const u64 *a, const u64 *b, u64 *out3, int nl, size_t count);
int keyswitch_generic(abc_hip_ctx *c, const u64 *target, size_t target_stride, const u64 *key, u64 *out2, int nl,
                      size_t count, const u64 *addend, size_t addend_stride, bool add_c1, u32 ginv = 0);
int launch_ks_tmod(abc_hip_ctx *c, const u64 *prodS, u64 *tmod, int nl, size_t polys);
int launch_ks_finish(abc_hip_ctx *c, const u64 *prodD, const u64 *tmod, u64 *out, const u64 *addend, size_t addend_stride,
                     bool add_c1, int nl, size_t count);
int launch_galois(abc_hip_ctx *c, const u64 *in, u64 *out, int nl, size_t polys, uint32_t elt, bool ntt_form);
int launch_rescale(abc_hip_ctx *c, const u64 *in, u64 *out, int size, int nl, size_t count);
int launch_drop_last(abc_hip_ctx *c, const u64 *in, u64 *out, int size, int nl, size_t count);

int bfv_multiply(abc_hip_ctx *c, const u64 *a, const u64 *b, u64 *out3, size_t count);
int bfv_multiply_plain(abc_hip_ctx *c, const u64 *ct, const u64 *plain, size_t plain_stride, u64 *out, int size, size_t count);
int bfv_addsub_plain(abc_hip_ctx *c, const u64 *ct, const u64 *plain, size_t plain_stride, u64 *out, int size, size_t count, int sub);
int ckks_multiply_plain(abc_hip_ctx *c, const u64 *ct, const u64 *plain, size_t plain_stride, u64 *out, int size, int nl, size_t count);
int ckks_add_plain(abc_hip_ctx *c, const u64 *ct, const u64 *plain, size_t plain_stride, u64 *out, int size, int nl, size_t count, int sub);
int batch_encode(abc_hip_ctx *c, const int64_t *values, u64 *plain, size_t count);
int batch_decode(abc_hip_ctx *c, const u64 *plain, int64_t *values, size_t count);
int encrypt(abc_hip_ctx *c, const u64 *plain, uint64_t seed, u64 *ct, size_t count);
int decrypt(abc_hip_ctx *c, const u64 *ct, int size, int nl, u64 *plain, size_t count);
int keygen(abc_hip_ctx *c, uint64_t seed);
int keygen_secure(abc_hip_ctx *c);
int encrypt_secure(abc_hip_ctx *c, const u64 *plain, u64 *ct, size_t count);
int microbench(abc_hip_ctx *c, int which, int iters, double *ms);

// LDS-resident fast paths (N <= 2^14); return -1 if not applicable (caller falls back to the generic kernels)
int rotate_fused(abc_hip_ctx *c, const u64 *in, u32 elt, const u64 *key, u64 *out, int nl, size_t count);
int keyswitch_fused(abc_hip_ctx *c, const u64 *target, size_t target_stride, const u64 *key, u64 *out2, int nl, size_t count,
                    const u64 *addend, size_t addend_stride, bool add_c1);
int ckks_mul_relin_fused(abc_hip_ctx *c, const u64 *a, const u64 *b, u64 *out, int nl, size_t count);
// split key switch without LDS-resident limbs (abc_kernels_gsplit.hip): N = 2^15, and the first step at N = 2^14 for small batches
bool gsplit_applies(const abc_hip_ctx *c, int nl);
size_t gsplit_scratch_words(const abc_hip_ctx *c, int nl);
int gsplit_chunk15(abc_hip_ctx *c, hipStream_t st, u64 *scratch, size_t cc, int nl, int mode, const u64 *opa, const u64 *opb,
                   size_t opa_stride, size_t opb_stride, int add_c1, const u64 *key, u64 *out, u32 gelt);
void gsplit_front14(hipStream_t st, abc_hip_ctx *c, size_t cc, int nl, int mode, const u64 *opa, const u64 *opb, size_t opa_stride,
                    double *hinv, double *part, u32 gelt, int pack = 0);
bool split4_main_subset(hipStream_t st, abc_hip_ctx *c, size_t cc, int nl, int mode, const double *part, const double *tpart, const u64 *opa,
                        const u64 *opb, size_t opa_stride, size_t opb_stride, int add_c1, const u64 *key, u64 *out, u32 gelt, u32 imap,
                        int ni);
bool gsplit_main_subset15(hipStream_t st, abc_hip_ctx *c, size_t cc, int nl, int mode, const double *part, const double *tpart, const u64 *opa,
                          const u64 *opb, size_t opa_stride, size_t opb_stride, int add_c1, const u64 *key, u64 *out, u32 gelt, u32 imap,
                          int ni);
void gsplit_main_deep_subset15(hipStream_t st, abc_hip_ctx *c, size_t cc, int nl, int mode, const double *part, const double *tpart,
                               const u64 *opa, const u64 *opb, size_t opa_stride, size_t opb_stride, int add_c1, const u64 *key, u64 *out,
                               u32 gelt, u64 imap, int ni);
bool bsplit_applies(const abc_hip_ctx *c, int nl);
bool bsplit_big_applies(const abc_hip_ctx *c, int nl);
bool iks_bfv_applies(const abc_hip_ctx *c, int nl);  // abc_kernels_eval.hip
int bsplit_big(abc_hip_ctx *c, const u64 *target, size_t target_stride, const u64 *key, u64 *out, int nl, size_t count, const u64 *addend,
               size_t addend_stride, bool add_c1, u32 ginv = 0);
int bsplit_back14(abc_hip_ctx *c, hipStream_t st, size_t cc, int nl, const double *part, double *half, const u64 *key, const u64 *addend,
                  size_t addend_stride, int add_c1, u64 *out, u32 ginv = 0);
int bsplit_back13(abc_hip_ctx *c, hipStream_t st, size_t cc, int nl, const double *part, double *half, const u64 *key, const u64 *addend,
                  size_t addend_stride, int add_c1, u64 *out, u32 ginv = 0);
// the fp64 twin of a key-switching key (nullptr: not available -- capture in progress and not built yet, or allocation failed)
const double *key_twin(abc_hip_ctx *c, const u64 *key);
const double *key_twin_lookup(const abc_hip_ctx *c, const u64 *key);  // never builds: safe once the lanes have forked
void drop_key_twins(abc_hip_ctx *c, const u64 *key /* nullptr: all */);
// internal lanes (streams forked off the context's stream): chunks of one call alternate over them (abc_kernels_fused.hip)
int fork_lanes(abc_hip_ctx *c, int lanes);
int join_lanes(abc_hip_ctx *c, int lanes);
// fork on construction (fork()), join on every way out: an early `return 1` between the two would otherwise leave work on
// the lanes that the context's stream -- and with it every later use or release of the buffers involved -- never waits for
struct LaneScope {
  abc_hip_ctx *c;
  int lanes;
  bool forked = false;
  LaneScope(abc_hip_ctx *c_, int lanes_) : c(c_), lanes(lanes_) {}
  int fork() {
    if (fork_lanes(c, lanes)) return 1;
    forked = true;
    return 0;
  }
  int join() {
    forked = false;
    return join_lanes(c, lanes);
  }
  ~LaneScope() {
    if (forked) (void)join_lanes(c, lanes);
  }
};

// BFV multiply (+ relinearise) in split form, N = 2^14 (abc_kernels_bmul.hip)
bool bmul_applies(const abc_hip_ctx *c);           // multiply + relinearise in one sequence
bool bmul_multiply_applies(const abc_hip_ctx *c);  // the multiply alone (also N = 2^15 / 2^16)
int launch_ntt_fwd_block_part(abc_hip_ctx *c, u64 *d, const LimbMap &map, int nl, size_t total_limbs);  // abc_kernels_ntt.hip
int launch_bfv_tensor_inv_block(abc_hip_ctx *c, const u64 *a, const u64 *b, size_t ct_stride, u64 *d, const LimbMap &map, int nlm,
                                size_t count);  // abc_kernels_bfv.hip
int bmul_split(abc_hip_ctx *c, const u64 *a, const u64 *b, u64 *out, size_t count, bool relin);
// integer twins of the split kernels (abc_kernels_isplit.hip)
bool isplit_applies(const abc_hip_ctx *c, int nl);
size_t isplit_scratch_words(const abc_hip_ctx *c, int nl);
int isplit_chunk(abc_hip_ctx *c, hipStream_t st, u64 *scratch, size_t cc, int nl, int mode, const u64 *opa, const u64 *opb,
                 size_t opa_stride, size_t opb_stride, int add_c1, const u64 *key, u64 *out, u32 gelt);

}  // namespace abc
