// abc_context.hip -- context construction (device tables) and the extern "C" entry points of
// libabc_hip.so declared in include/abc_hip.h.
//
// abc_hip_ctx_create replaces SealCiphertextFactory::setupSealContext
// (src/runtime/SealCiphertextFactory.cpp:72-100): it fixes the modulus chain, builds the NTT twiddle
// tables (bit-reversed powers of the minimal primitive 2N-th root, with Shoup quotients), the BEHZ
// auxiliary bases {B, m_sk, gamma, m~ = 2^32} and every scalar constant the kernels need, and uploads
// them once; nothing is recomputed per operation.
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../../include/abc_hip.h"
#include <dlfcn.h>

#include "abc_context.hpp"
#include "abc_host_math.hpp"

namespace abc {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }

static bool env_on(const char *name) {  // set and not "0"
  const char *e = std::getenv(name);
  return e && !(e[0] == '0' && e[1] == 0);
}
void read_switches(abc_hip_ctx *c) {
  abc_hip_ctx::Switches s;
  s.no_fused = env_on("ABC_HIP_NO_FUSED");
  s.no_split = env_on("ABC_HIP_NO_SPLIT");
  s.no_split4 = env_on("ABC_HIP_NO_SPLIT4");
  s.no_isplit = env_on("ABC_HIP_NO_ISPLIT");
  s.no_gsplit = env_on("ABC_HIP_NO_GSPLIT");
  s.no_bsplit = env_on("ABC_HIP_NO_BSPLIT");
  s.no_mixed = env_on("ABC_HIP_NO_MIXED");
  s.no_pack = env_on("ABC_HIP_NO_PACK");
  s.no_key_twin = env_on("ABC_HIP_NO_KEY_TWIN");
  s.no_bmul = env_on("ABC_HIP_NO_BMUL");
  s.no_special8x2 = env_on("ABC_HIP_NO_SPECIAL8X2");
  s.no_bmul_mid = env_on("ABC_HIP_NO_BMUL_MID");
  s.no_finish_lds = env_on("ABC_HIP_NO_FINISH_LDS");
  s.no_iks = env_on("ABC_HIP_NO_IKS");
  s.no_bmul_r6 = env_on("ABC_HIP_NO_BMUL_R6");
  s.no_lean_front = env_on("ABC_HIP_NO_LEAN_FRONT");
  s.no_tensor_intt = env_on("ABC_HIP_NO_TENSOR_INTT");
  s.no_galois_fusion = env_on("ABC_HIP_NO_GALOIS_FUSION");
  if (const char *e = std::getenv("ABC_HIP_CHUNK")) s.chunk = (size_t)std::atol(e);
  if (const char *e = std::getenv("ABC_HIP_FEW_LIMBS")) s.few_limbs = (size_t)std::atol(e);
  if (const char *e = std::getenv("ABC_HIP_LEAN_LIMIT")) s.lean_limit = (size_t)std::atol(e);
  if (const char *e = std::getenv("ABC_HIP_PASS0_TARGET_LIMIT")) s.pass0_target_limit = (size_t)std::atol(e);
  if (const char *e = std::getenv("ABC_HIP_BFV_SCRATCH_MB")) s.bfv_scratch_mb = (size_t)std::atol(e);
  if (const char *e = std::getenv("ABC_HIP_LANE_OFFSET_US")) s.lane_offset_us = (unsigned)std::atoi(e);
  if (const char *e = std::getenv("ABC_HIP_LANES")) {
    s.lanes = std::atoi(e);
    if (s.lanes < 1) s.lanes = 1;
    if (s.lanes > abc_hip_ctx::kMaxLanes) s.lanes = abc_hip_ctx::kMaxLanes;
  }
  c->sw = s;
}

static bool capturing(abc_hip_ctx *c) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(c->stream, &st) != hipSuccess) return false;
  return st != hipStreamCaptureStatusNone;
}

// Give every cached (free-listed) block back to the driver.  The stream is drained first: a cached block may still be
// read by work that was enqueued before it was freed.
static int trim_cache(abc_hip_ctx *c) {
  std::lock_guard<std::mutex> lock(c->alloc_mu);
  if (c->free_blocks.empty()) return 0;
  ABC_HIP_CHECK(hipStreamSynchronize(c->stream));
  for (auto &kv : c->free_blocks)
    for (void *p : kv.second) {
      c->block_size.erase(p);
      (void)hipFree(p);
    }
  c->free_blocks.clear();
  c->cached_bytes = 0;
  return 0;
}
// hipMalloc; on failure flush this context's cache once and try again (the cache never shrinks by itself)
static hipError_t malloc_retry(abc_hip_ctx *c, void **p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipSuccess) return e;
  (void)hipGetLastError();
  if (trim_cache(c)) return e;
  return hipMalloc(p, bytes);
}

int ensure_workspace(abc_hip_ctx *c, size_t bytes) {
  if (bytes <= c->ws_bytes) return 0;
  if (capturing(c)) { set_error("scratch would grow during graph capture: run the sequence once eagerly first"); return 1; }
  if (c->ws) {
    ABC_HIP_CHECK(hipStreamSynchronize(c->stream));
    (void)hipFree(c->ws);
    c->ws = nullptr;
    c->ws_bytes = 0;
  }
  size_t want = bytes + bytes / 8;
  ABC_HIP_CHECK(malloc_retry(c, &c->ws, want));
  c->ws_bytes = want;
  return 0;
}

int ensure_aux(abc_hip_ctx *c, int which, size_t bytes) {
  if (bytes <= c->aux_bytes[which]) return 0;
  if (capturing(c)) { set_error("scratch would grow during graph capture: run the sequence once eagerly first"); return 1; }
  if (c->aux[which]) {
    ABC_HIP_CHECK(hipStreamSynchronize(c->stream));
    (void)hipFree(c->aux[which]);
    c->aux[which] = nullptr;
    c->aux_bytes[which] = 0;
  }
  ABC_HIP_CHECK(malloc_retry(c, &c->aux[which], bytes));
  c->aux_bytes[which] = bytes;
  return 0;
}

static Mod make_mod(uint64_t q, int logn, bool ntt) {
  using namespace host;
  Mod m{};
  m.q = q;
  m.two_q = 2 * q;
  const int k = bitlen(q);
  m.bits = (u32)k;
  m.shift = (u32)(k - 1);
  m.mu = (uint64_t)((((u128)1) << (k + 63)) / q);
  m.qd = (double)q;
  m.qinv = 1.0 / (double)q;
  if (ntt) {
    const uint64_t n = 1ull << logn;
    m.inv_n = invmod(n % q, q);
    m.inv_n_s = shoup(m.inv_n, q);
    if (fp_ok(m.bits)) {
      m.inv_n_c = m.inv_n > q / 2 ? -(double)(q - m.inv_n) : (double)m.inv_n;
      m.inv_n_cq = m.inv_n_c / m.qd;
    }
  }
  return m;
}

static void fill_twiddles(uint64_t q, int logn, uint64_t *dst /*[2][N][2]*/) {
  using namespace host;
  const size_t n = (size_t)1 << logn;
  const uint64_t psi = min_primitive_root(2 * n, q);
  uint64_t *fwd = dst, *inv = dst + 2 * n;  // interleaved {w, Shoup quotient}
  uint64_t p = 1;
  for (size_t i = 0; i < n; i++) {
    fwd[2 * (size_t)bitrev((uint32_t)i, logn)] = p;
    p = mulmod(p, psi, q);
  }
  const uint64_t ipsi = invmod(psi, q);
  p = 1;
  for (size_t i = 0; i < n; i++) {
    inv[2 * (size_t)bitrev((uint32_t)i, logn)] = p;
    p = mulmod(p, ipsi, q);
  }
  for (size_t i = 0; i < n; i++) {
    fwd[2 * i + 1] = shoup(fwd[2 * i], q);
    inv[2 * i + 1] = shoup(inv[2 * i], q);
  }
}

// fp64 twin of one modulus' twiddle table: {w centred into (-q/2, q/2], w / q} (both exact-operand roundings)
static void fill_fp_twiddles(uint64_t q, size_t n, const uint64_t *src /*[2][N][2]*/, double *dst) {
  const double qd = (double)q;
  for (size_t i = 0; i < 2 * n; i++) {
    const uint64_t w = src[2 * i];
    const double wc = w > q / 2 ? -(double)(q - w) : (double)w;
    dst[2 * i] = wc;
    dst[2 * i + 1] = wc / qd;
  }
}

static int build_context(abc_hip_ctx *c) {
  using namespace host;
  const int logn = c->logn, K = c->K, L = c->L;
  const size_t N = (size_t)c->n;
  const bool bfv = (c->scheme == ABC_HIP_SCHEME_BFV);
  std::vector<uint64_t> qs(c->primes.begin(), c->primes.begin() + L);
  const uint64_t qsp = c->primes[K - 1];
  DevConst &k = c->h_cst;
  std::memset(&k, 0, sizeof(k));

  // ---- modulus table ----
  c->mod_values.assign(c->primes.begin(), c->primes.end());
  std::vector<uint64_t> Bp;
  uint64_t m_sk = 0, gamma = 0;
  if (bfv) {
    // BEHZ auxiliary base, sized by SEAL's RNSTool rule.  SEAL draws m_sk, gamma and B from the 61-bit NTT primes.  The
    // q-residues a BFV multiply returns do not depend on which auxiliary primes carry the intermediate values (every
    // step computes residues of one fixed integer -- DESIGN.md section 4c; shown on the oracle by
    // tests/test_oracle_internal.py and on the device by every BFV parity test, whose oracle keeps SEAL's base), so where
    // every ciphertext prime is below 2^50 the base is drawn from the 50-bit primes instead: all BEHZ transforms then
    // take the exact-fp64 butterflies (8 instead of 16 instructions).  gamma (decryption only) stays SEAL's.
    bool fp_aux = !env_on("ABC_HIP_NO_FP64") && !env_on("ABC_HIP_BEHZ_SEAL_BASE");
    for (uint64_t q : c->primes) fp_aux = fp_aux && fp_ok((u32)bitlen(q));
    const int aux_bits = fp_aux ? 50 : 61;
    int nB = L;
    if (32 + bitlen(c->t) + prod_bitlen(qs) >= aux_bits * L + aux_bits) nB++;
    gamma = ntt_primes(N, 61, 2)[1];
    if (!fp_aux) {
      auto aux = ntt_primes(N, 61, (size_t)nB + 2);
      m_sk = aux[0];
      Bp.assign(aux.begin() + 2, aux.end());
    } else {
      std::vector<uint64_t> alt;
      for (uint64_t p : ntt_primes(N, 50, (size_t)nB + 1 + (size_t)K)) {
        bool clash = false;
        for (uint64_t q : c->primes) clash = clash || (p == q);
        if (!clash && alt.size() < (size_t)nB + 1) alt.push_back(p);
      }
      if (alt.size() < (size_t)nB + 1) throw std::runtime_error("not enough 50-bit NTT primes for the BEHZ auxiliary base");
      m_sk = alt[0];
      Bp.assign(alt.begin() + 1, alt.end());
    }
    // sums of up to L + 1 (resp. nB + 1) lazily reduced terms must stay below 2^53: 12 x 0.625 x 2^50
    c->behz_fp = fp_aux && L <= 10 && !env_on("ABC_HIP_BEHZ_INT_KERNELS");
    c->nB = nB;
    c->nBsk = nB + 1;
    for (uint64_t b : Bp) c->mod_values.push_back(b);
    c->mod_values.push_back(m_sk);
    c->mod_values.push_back(gamma);
    c->mod_values.push_back(c->t);
  }
  const int nmods = (int)c->mod_values.size();
  const int id_bsk = K, id_gamma = K + c->nBsk, id_t = K + c->nBsk + 1;
  c->h_mods.clear();
  std::vector<uint64_t> h_tw((size_t)nmods * 4 * N, 0);
  std::vector<double> h_ftw((size_t)nmods * 4 * N, 0.0);
  for (int i = 0; i < nmods; i++) {
    const bool ntt = !(bfv && i == id_gamma);
    c->h_mods.push_back(make_mod(c->mod_values[i], logn, ntt));
    if (ntt) fill_twiddles(c->mod_values[i], logn, h_tw.data() + (size_t)i * 4 * N);
    if (ntt && fp_ok(c->h_mods.back().bits))
      fill_fp_twiddles(c->mod_values[i], N, h_tw.data() + (size_t)i * 4 * N, h_ftw.data() + (size_t)i * 4 * N);
  }

  // ---- key / modulus switching constants ----
  for (int j = 0; j < L; j++) {
    const uint64_t q = qs[j];
    k.inv_special[j] = invmod(qsp % q, q);
    k.inv_special_s[j] = shoup(k.inv_special[j], q);
    k.special_mod_q[j] = qsp % q;
    k.special_mod_q_s[j] = shoup(k.special_mod_q[j], q);
    k.inv_special_c[j] = k.inv_special[j] > q / 2 ? -(double)(q - k.inv_special[j]) : (double)k.inv_special[j];
    k.inv_special_cq[j] = k.inv_special_c[j] / (double)q;
    k.special_c[j] = k.special_mod_q[j] > q / 2 ? -(double)(q - k.special_mod_q[j]) : (double)k.special_mod_q[j];
    k.special_cq[j] = k.special_c[j] / (double)q;
  }
  for (int l = 1; l < L; l++)
    for (int j = 0; j < l; j++) {
      k.inv_qlast[l][j] = invmod(qs[l] % qs[j], qs[j]);
      k.inv_qlast_s[l][j] = shoup(k.inv_qlast[l][j], qs[j]);
      k.inv_qlast_c[l][j] = k.inv_qlast[l][j] > qs[j] / 2 ? -(double)(qs[j] - k.inv_qlast[l][j]) : (double)k.inv_qlast[l][j];
      k.inv_qlast_cq[l][j] = k.inv_qlast_c[l][j] / (double)qs[j];
    }

  std::vector<uint32_t> slot_map;
  if (bfv) {
    const uint64_t t = c->t;
    k.t = t;
    k.q_mod_t = prod_mod(qs, t);
    k.upper_half_threshold = (t + 1) >> 1;
    for (int i = 0; i < L; i++) {
      const uint64_t q = qs[i];
      // floor(Q/t) mod q_i = -(Q mod t) * t^-1 mod q_i
      k.coeff_div_plain[i] = mulmod(negmod(k.q_mod_t, q), invmod(t % q, q), q);
      k.upper_half_increment[i] = q - t;
    }
    // BatchEncoder slot -> coefficient-index map: slot i of row 0 <-> evaluation at psi^(3^i)
    slot_map.resize(N);
    const size_t row = N >> 1, m = N << 1;
    uint64_t pos = 1;
    for (size_t i = 0; i < row; i++) {
      slot_map[i] = bitrev((uint32_t)((pos - 1) >> 1), logn);
      slot_map[row | i] = bitrev((uint32_t)((m - pos - 1) >> 1), logn);
      pos = (pos * 3) & (m - 1);
    }
    // ---- BEHZ ----
    const int nB = c->nB, nBsk = c->nBsk;
    const uint64_t mt = 1ull << 32;
    std::vector<uint64_t> bsk(Bp);
    bsk.push_back(m_sk);
    k.nq = L; k.nB = nB; k.nBsk = nBsk;
    auto punct_mod = [&](const std::vector<uint64_t> &base, int skip, uint64_t p) {
      uint64_t v = 1 % p;
      for (int i = 0; i < (int)base.size(); i++)
        if (i != skip) v = mulmod(v, base[i] % p, p);
      return v;
    };
    for (int i = 0; i < L; i++) {
      const uint64_t q = qs[i];
      k.mtilde_mod_q[i] = mt % q;
      k.inv_punct_q[i] = invmod(punct_mod(qs, i, q), q);
      for (int j = 0; j < nBsk; j++) {
        k.q_to_bsk[j][i] = punct_mod(qs, i, bsk[j]);
        k.q_to_bsk_s[j][i] = shoup(k.q_to_bsk[j][i], bsk[j]);
      }
      k.q_to_mtilde[i] = punct_mod(qs, i, mt);
      k.q_to_t[i] = punct_mod(qs, i, t);
      k.q_to_gamma[i] = punct_mod(qs, i, gamma);
      k.B_mod_q[i] = prod_mod(Bp, q);
      k.t_mod_q[i] = t % q;
      k.tgamma_mod_q[i] = mulmod(t % q, gamma % q, q);
      for (int b = 0; b < nB; b++) {
        k.B_to_q[i][b] = punct_mod(Bp, b, q);
        k.B_to_q_s[i][b] = shoup(k.B_to_q[i][b], q);
      }
    }
    k.neg_inv_q_mod_mtilde = negmod(invmod(prod_mod(qs, mt), mt), mt);
    for (int j = 0; j < nBsk; j++) {
      const uint64_t p = bsk[j];
      k.q_mod_bsk[j] = prod_mod(qs, p);
      k.q_mod_bsk_s[j] = shoup(k.q_mod_bsk[j], p);
      k.inv_q_mod_bsk[j] = invmod(k.q_mod_bsk[j], p);
      k.inv_mtilde_mod_bsk[j] = invmod(mt % p, p);
      k.t_mod_bsk[j] = t % p;
    }
    for (int b = 0; b < nB; b++) {
      k.inv_punct_B[b] = invmod(punct_mod(Bp, b, Bp[b]), Bp[b]);
      k.B_to_msk[b] = punct_mod(Bp, b, m_sk);
      k.B_to_msk_s[b] = shoup(k.B_to_msk[b], m_sk);
    }
    k.inv_B_mod_msk = invmod(prod_mod(Bp, m_sk), m_sk);
    k.inv_B_mod_msk_s = shoup(k.inv_B_mod_msk, m_sk);
    for (int i = 0; i < L; i++) {
      const uint64_t q = qs[i];
      k.ext_q[i] = mulmod(k.mtilde_mod_q[i], k.inv_punct_q[i], q);
      k.ext_q_s[i] = shoup(k.ext_q[i], q);
      k.flr_q[i] = mulmod(k.t_mod_q[i], k.inv_punct_q[i], q);
      k.flr_q_s[i] = shoup(k.flr_q[i], q);
      k.dec_q[i] = mulmod(k.tgamma_mod_q[i], k.inv_punct_q[i], q);
      k.dec_q_s[i] = shoup(k.dec_q[i], q);
      k.B_mod_q_s[i] = shoup(k.B_mod_q[i], q);
    }
    for (int j = 0; j < nBsk; j++) {
      const uint64_t p = bsk[j];
      k.tinvq_bsk[j] = mulmod(k.t_mod_bsk[j], k.inv_q_mod_bsk[j], p);
      k.tinvq_bsk_s[j] = shoup(k.tinvq_bsk[j], p);
      k.inv_q_mod_bsk_s[j] = shoup(k.inv_q_mod_bsk[j], p);
      k.inv_mtilde_mod_bsk_s[j] = shoup(k.inv_mtilde_mod_bsk[j], p);
    }
    for (int b = 0; b < nB; b++) k.inv_punct_B_s[b] = shoup(k.inv_punct_B[b], Bp[b]);
    k.neg_inv_q_mod_t = negmod(invmod(prod_mod(qs, t), t), t);
    k.neg_inv_q_mod_gamma = negmod(invmod(prod_mod(qs, gamma), gamma), gamma);
    k.inv_gamma_mod_t = invmod(gamma % t, t);
    k.gamma = gamma;
  }

  // ---- upload ----
  ABC_HIP_CHECK(hipMalloc(&c->d_mods, nmods * sizeof(Mod)));
  ABC_HIP_CHECK(hipMemcpy(c->d_mods, c->h_mods.data(), nmods * sizeof(Mod), hipMemcpyHostToDevice));
  ABC_HIP_CHECK(hipMalloc(&c->d_tw, h_tw.size() * 8));
  ABC_HIP_CHECK(hipMemcpy(c->d_tw, h_tw.data(), h_tw.size() * 8, hipMemcpyHostToDevice));
  ABC_HIP_CHECK(hipMalloc(&c->d_ftw, h_ftw.size() * 8));
  ABC_HIP_CHECK(hipMemcpy(c->d_ftw, h_ftw.data(), h_ftw.size() * 8, hipMemcpyHostToDevice));
  c->use_fp = !env_on("ABC_HIP_NO_FP64");
  read_switches(c);
  ABC_HIP_CHECK(hipMalloc(&c->d_cst, sizeof(DevConst)));
  ABC_HIP_CHECK(hipMemcpy(c->d_cst, &k, sizeof(DevConst), hipMemcpyHostToDevice));
  if (bfv && c->behz_fp) {
    // {centred value, value / p} pairs of every BEHZ constant
    auto pair = [](double (&dst)[2], uint64_t v, uint64_t p) {
      const double vc = v > p / 2 ? -(double)(p - v) : (double)v;
      dst[0] = vc;
      dst[1] = vc / (double)p;
    };
    std::vector<DevConstFp> hf(1);
    DevConstFp &f = hf[0];
    std::memset(&f, 0, sizeof(f));
    const int nB = c->nB, nBsk = c->nBsk;
    std::vector<uint64_t> bskv(Bp);
    bskv.push_back(m_sk);
    for (int i = 0; i < L; i++) {
      pair(f.ext_q[i], k.ext_q[i], qs[i]);
      pair(f.flr_q[i], k.flr_q[i], qs[i]);
      pair(f.B_mod_q[i], k.B_mod_q[i], qs[i]);
      for (int b = 0; b < nB; b++) pair(f.B_to_q[i][b], k.B_to_q[i][b], qs[i]);
    }
    for (int j = 0; j < nBsk; j++) {
      for (int i = 0; i < L; i++) pair(f.q_to_bsk[j][i], k.q_to_bsk[j][i], bskv[j]);
      pair(f.q_mod_bsk[j], k.q_mod_bsk[j], bskv[j]);
      pair(f.inv_mtilde_mod_bsk[j], k.inv_mtilde_mod_bsk[j], bskv[j]);
      pair(f.tinvq_bsk[j], k.tinvq_bsk[j], bskv[j]);
      pair(f.inv_q_mod_bsk[j], k.inv_q_mod_bsk[j], bskv[j]);
    }
    for (int b = 0; b < nB; b++) {
      pair(f.inv_punct_B[b], k.inv_punct_B[b], Bp[b]);
      pair(f.B_to_msk[b], k.B_to_msk[b], m_sk);
    }
    pair(f.inv_B_mod_msk, k.inv_B_mod_msk, m_sk);
    ABC_HIP_CHECK(hipMalloc(&c->d_cstf, sizeof(DevConstFp)));
    ABC_HIP_CHECK(hipMemcpy(c->d_cstf, &f, sizeof(DevConstFp), hipMemcpyHostToDevice));
  }
  if (bfv) {
    ABC_HIP_CHECK(hipMalloc(&c->d_slot_map, N * 4));
    ABC_HIP_CHECK(hipMemcpy(c->d_slot_map, slot_map.data(), N * 4, hipMemcpyHostToDevice));
  }
  DevCtx &dc = c->dc;
  dc.mods = c->d_mods; dc.tw = c->d_tw; dc.ftw = c->d_ftw; dc.cst = c->d_cst; dc.cstf = c->d_cstf; dc.slot_map = c->d_slot_map;
  dc.logn = logn; dc.n = (int)N; dc.K = K; dc.L = L;
  dc.ps = (int)N;
  if (const char *e = std::getenv("ABC_HIP_SCRATCH_PAD")) dc.ps = (int)N + (std::atoi(e) / 2) * 2;  // words, kept even (16-byte rows)
  dc.id_bsk = id_bsk; dc.id_t = id_t; dc.id_gamma = id_gamma; dc.id_mtilde = -1;
  return 0;
}

// ---- rotation bookkeeping (GaloisTool::get_elt_from_step, util::naf) ----
static uint32_t elt_from_step(const abc_hip_ctx *c, int step) {
  const uint32_t n = (uint32_t)c->n;
  const uint64_t m = 2ull * n;
  if (step == 0) return (uint32_t)(m - 1);
  const bool neg = step < 0;
  const uint32_t pos = (uint32_t)(neg ? -step : step);
  if (pos >= (n >> 1)) return 0;
  const int s = neg ? (int)(n >> 1) - (int)pos : (int)pos;
  uint64_t g = 1;
  for (int i = 0; i < s; i++) g = (g * 3) & (m - 1);
  return (uint32_t)g;
}

static std::vector<int> naf(int value) {
  std::vector<int> out;
  const bool sign = value < 0;
  value = std::abs(value);
  for (int i = 0; value; i++) {
    const int zi = (value & 1) ? 2 - (value & 3) : 0;
    value = (value - zi) >> 1;
    if (zi) out.push_back((sign ? -zi : zi) * (1 << i));
  }
  return out;
}

// key switch dispatcher: LDS-resident kernels when the ring fits, generic kernels otherwise
static int keyswitch(abc_hip_ctx *c, const u64 *target, size_t target_stride, const u64 *key, u64 *out2, int nl, size_t count,
                     const u64 *addend, size_t addend_stride, bool add_c1) {
  const int rc = keyswitch_fused(c, target, target_stride, key, out2, nl, count, addend, addend_stride, add_c1);
  if (rc >= 0) return rc;
  return keyswitch_generic(c, target, target_stride, key, out2, nl, count, addend, addend_stride, add_c1);
}

static int apply_galois(abc_hip_ctx *c, const u64 *in, u64 *out, int nl, uint32_t elt, size_t count) {
  auto it = c->d_galois.find(elt);
  if (it == c->d_galois.end()) { set_error("Galois key not present"); return 1; }
  if (!count) return 0;
  const size_t N = (size_t)c->n, pw = (size_t)nl * N;
  const bool ntt_form = (c->scheme == ABC_HIP_SCHEME_CKKS);
  {  // N = 2^14 CKKS: the permutation is a block-local gather, done inside the key-switch kernels
    const int rc = rotate_fused(c, in, elt, it->second, out, nl, count);
    if (rc >= 0) return rc;
  }
  // g(c0), g(c1) live in arena 0 (keyswitch_generic uses c->ws)
  if (ensure_aux(c, 0, count * 2 * pw * 8)) return 1;
  u64 *g = (u64 *)c->aux[0];
  if (launch_galois(c, in, g, nl, count * 2, elt, ntt_form)) return 1;
  // out = (g(c0) + ks0, ks1) with ks = KeySwitch(g(c1))
  return keyswitch(c, g + pw, 2 * pw, it->second, out, nl, count, g, 2 * pw, false);
}

static int rotate(abc_hip_ctx *c, const u64 *in, u64 *out, int nl, int steps, size_t count) {
  const size_t bytes = count * 2 * nl * (size_t)c->n * 8;
  if (steps == 0) {
    if (in != out) ABC_HIP_CHECK(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, c->stream));
    return 0;
  }
  const uint32_t elt = elt_from_step(c, steps);
  if (!elt) { set_error("step count too large"); return 1; }
  std::vector<int> terms;
  if (c->d_galois.count(elt)) {
    terms.push_back(steps);
  } else {
    // no key for this element: SEAL decomposes the step count in non-adjacent form (Evaluator::rotate_internal)
    for (int s : naf(steps))
      if ((size_t)std::abs(s) != ((size_t)c->n >> 1)) terms.push_back(s);
    if (naf(steps).size() == 1) { set_error("Galois key not present"); return 1; }
    for (int s : terms)
      if (!c->d_galois.count(elt_from_step(c, s))) { set_error("Galois key not present"); return 1; }
  }
  // apply_galois is out of place: ping-pong through arenas 1 and 2, last hop lands in `out`
  const u64 *cur = in;
  for (size_t i = 0; i < terms.size(); i++) {
    const bool last = (i + 1 == terms.size());
    u64 *dst = out;
    if (!last || out == cur) {
      const int which = 1 + (int)(i & 1);
      if (ensure_aux(c, which, bytes)) return 1;
      dst = (u64 *)c->aux[which];
    }
    if (apply_galois(c, cur, dst, nl, elt_from_step(c, terms[i]), count)) return 1;
    cur = dst;
  }
  if (cur != out) ABC_HIP_CHECK(hipMemcpyAsync(out, cur, bytes, hipMemcpyDeviceToDevice, c->stream));
  return 0;
}

static int relinearize(abc_hip_ctx *c, const u64 *ct3, u64 *out2, int nl, size_t count) {
  if (!c->d_relin) { set_error("relinearize: no relinearisation key"); return 1; }
  const size_t pw = (size_t)nl * c->n;
  // target = c2 (poly 2 of each size-3 ciphertext); addend = (c0, c1)
  return keyswitch(c, ct3 + 2 * pw, 3 * pw, c->d_relin, out2, nl, count, ct3, 3 * pw, true);
}

}  // namespace abc

using namespace abc;

// roctx range around every C-ABI operation (rocprofv3 --marker-trace then shows the op boundaries above the kernel rows; the
// reference's only instrumentation is four wall-clock phase timers, examples/main.cpp:41).  The marker library is bound with
// dlopen on first use -- libabc_hip.so carries no link-time dependency on a profiler -- and ABC_HIP_NO_ROCTX=1 skips it.
namespace abc {
struct Roctx {
  int (*push)(const char *) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    if (env_on("ABC_HIP_NO_ROCTX")) return;
    for (const char *lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
      if (void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL)) {
        push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
        pop = (int (*)())dlsym(h, "roctxRangePop");
        if (push && pop) return;
        push = nullptr; pop = nullptr;
      }
    }
  }
};
static const Roctx &roctx() {
  static const Roctx r;
  return r;
}
struct OpRange {
  bool on;
  explicit OpRange(const char *name) : on(roctx().push != nullptr) {
    if (on) roctx().push(name);
  }
  ~OpRange() {
    if (on) roctx().pop();
  }
};
}  // namespace abc

#define CTX_GUARD(c)                                   \
  abc::OpRange abc_op_range_(__func__);                \
  do {                                                 \
    if (!(c)) { set_error("null context"); return 1; } \
    if (hipSetDevice((c)->device) != hipSuccess) { set_error("hipSetDevice failed"); return 1; } \
  } while (0)

// Host-synchronising entry points cannot be recorded, and letting HIP find that out invalidates the capture for good on this
// runtime (the stream keeps returning hipErrorStreamCaptureInvalidated even after hipStreamEndCapture): refuse up front.
#define NOT_CAPTURABLE(c, what)                                                                                   \
  do {                                                                                                            \
    if ((c)->capture_active) { set_error(what ": not capturable (host transfer / synchronisation inside abc_hip_graph_begin..end)"); return 1; } \
  } while (0)

extern "C" {

const char *abc_hip_last_error(void) { return g_err.c_str(); }

int abc_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int abc_hip_default_bfv_primes(size_t n, uint64_t *out) {
  // the 128-bit-security default chains of SEAL 3.6 (CoeffModulus::BFVDefault)
  static const uint64_t p1024[] = {0x7e00001ull};
  static const uint64_t p2048[] = {0x3fffffff000001ull};
  static const uint64_t p4096[] = {0xffffee001ull, 0xffffc4001ull, 0x1ffffe0001ull};
  static const uint64_t p8192[] = {0x7fffffd8001ull, 0x7fffffc8001ull, 0xfffffffc001ull, 0xffffff6c001ull, 0xfffffebc001ull};
  static const uint64_t p16384[] = {0xfffffffd8001ull,  0xfffffffa0001ull,  0xfffffff00001ull,  0x1fffffff68001ull, 0x1fffffff50001ull,
                                    0x1ffffffee8001ull, 0x1ffffffea0001ull, 0x1ffffffe88001ull, 0x1ffffffe48001ull};
  static const uint64_t p32768[] = {0x7fffffffe90001ull, 0x7fffffffbf0001ull, 0x7fffffffbd0001ull, 0x7fffffffba0001ull,
                                    0x7fffffffaa0001ull, 0x7fffffffa50001ull, 0x7fffffff9f0001ull, 0x7fffffff7e0001ull,
                                    0x7fffffff770001ull, 0x7fffffff380001ull, 0x7fffffff330001ull, 0x7fffffff2d0001ull,
                                    0x7fffffff170001ull, 0x7fffffff150001ull, 0x7ffffffef00001ull, 0xfffffffff70001ull};
  const uint64_t *src = nullptr;
  int cnt = 0;
  switch (n) {
    case 1024: src = p1024; cnt = 1; break;
    case 2048: src = p2048; cnt = 1; break;
    case 4096: src = p4096; cnt = 3; break;
    case 8192: src = p8192; cnt = 5; break;
    case 16384: src = p16384; cnt = 9; break;
    case 32768: src = p32768; cnt = 16; break;
    default: set_error("no default BFV coefficient modulus for this ring degree"); return -1;
  }
  for (int i = 0; i < cnt; i++) out[i] = src[i];
  return cnt;
}

uint64_t abc_hip_plain_modulus_batching(size_t n, int bits) {
  try {
    return host::ntt_primes(n, bits, 1)[0];
  } catch (const std::exception &e) {
    set_error(e.what());
    return 0;
  }
}

int abc_hip_create_primes(size_t n, const int *bit_sizes, int count, uint64_t *out) {
  try {
    std::map<int, std::vector<uint64_t>> table;
    for (int i = 0; i < count; i++) table[bit_sizes[i]];
    for (auto &kv : table) {
      size_t need = 0;
      for (int i = 0; i < count; i++) need += (bit_sizes[i] == kv.first);
      kv.second = host::ntt_primes(n, kv.first, need);
    }
    for (int i = 0; i < count; i++) {  // hand out from the back (smallest first), as CoeffModulus::Create does
      auto &v = table[bit_sizes[i]];
      out[i] = v.back();
      v.pop_back();
    }
    return 0;
  } catch (const std::exception &e) {
    set_error(e.what());
    return 1;
  }
}

int abc_hip_ctx_create(int scheme, int logn, const uint64_t *primes, int nprimes, uint64_t plain_modulus, int device,
                       abc_hip_ctx **out) {
  if (!out) { set_error("null output pointer"); return 1; }
  *out = nullptr;
  if (scheme != ABC_HIP_SCHEME_BFV && scheme != ABC_HIP_SCHEME_CKKS) { set_error("unknown scheme"); return 1; }
  if (logn < 10 || logn > 16) { set_error("logn must be in 10..16"); return 1; }
  if (nprimes < 2 || nprimes > kMaxLimbs) { set_error("need 2..16 primes (data limbs + special prime)"); return 1; }
  const uint64_t two_n = 2ull << logn;
  for (int i = 0; i < nprimes; i++) {
    if (primes[i] >> 60 || !host::is_prime(primes[i]) || primes[i] % two_n != 1) {
      set_error("coefficient modulus primes must be < 2^60, prime and = 1 mod 2N");
      return 1;
    }
    for (int j = 0; j < i; j++)
      if (primes[j] == primes[i]) { set_error("coefficient modulus primes must be distinct"); return 1; }
  }
  if (scheme == ABC_HIP_SCHEME_BFV) {
    if (!host::is_prime(plain_modulus) || plain_modulus % two_n != 1 || plain_modulus >> 32) {
      set_error("plain modulus must be a prime = 1 mod 2N (batching) below 2^32");
      return 1;
    }
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available"); return 1; }
  if (device < 0 || device >= ndev) { set_error("device index out of range"); return 1; }
  if (hipSetDevice(device) != hipSuccess) { set_error("hipSetDevice failed"); return 1; }
  abc_hip_ctx *c = new abc_hip_ctx();
  c->scheme = scheme; c->logn = logn; c->n = 1 << logn; c->K = nprimes; c->L = nprimes - 1; c->device = device;
  c->primes.assign(primes, primes + nprimes);
  c->t = (scheme == ABC_HIP_SCHEME_BFV) ? plain_modulus : 0;
  int rc = 1;
  try {
    rc = build_context(c);
  } catch (const std::exception &e) {
    set_error(e.what());
  }
  if (!rc && hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
    set_error("hipStreamCreate failed");
    rc = 1;
  }
  if (!rc) c->stream = c->own_stream;  // operations run on a private stream unless abc_hip_set_stream overrides it
  if (!rc && (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
              hipEventCreateWithFlags(&c->lane_fork, hipEventDisableTiming) != hipSuccess)) {
    set_error("hipEventCreate failed");
    rc = 1;
  }
  // ABC_HIP_CU_MASK=<m>: lane i may only use the CUs whose index is congruent to i modulo m (experiment: forces two
  // lanes to run side by side, a memory-bound kernel on one half of every XCD next to an arithmetic-bound one)
  int mask_mod = 0;
  if (const char *e = std::getenv("ABC_HIP_CU_MASK")) mask_mod = std::atoi(e);
  for (int i = 0; !rc && i < abc_hip_ctx::kMaxLanes; i++) {
    hipError_t se;
    if (mask_mod >= 2) {
      uint32_t mask[8];
      for (int w = 0; w < 8; w++) {
        mask[w] = 0;
        for (int b = 0; b < 32; b++)
          if (((w * 32 + b) % mask_mod) == (i % mask_mod)) mask[w] |= 1u << b;
      }
      se = hipExtStreamCreateWithCUMask(&c->lane[i], 8, mask);
    } else {
      se = hipStreamCreateWithFlags(&c->lane[i], hipStreamNonBlocking);
    }
    if (hipEventCreateWithFlags(&c->lane_join[i], hipEventDisableTiming) != hipSuccess || se != hipSuccess) {
      set_error("lane stream / event creation failed");
      rc = 1;
    }
  }
  if (!rc && !env_on("ABC_HIP_SYNC_ALLOC")) {
    c->cache_alloc = true;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) c->cache_cap = total_b / 4;  // at most a quarter of the device
  }
  if (rc) { abc_hip_ctx_destroy(c); return 1; }
  *out = c;
  return 0;
}

void abc_hip_ctx_destroy(abc_hip_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  (void)hipFree(c->d_mods); (void)hipFree(c->d_tw); (void)hipFree(c->d_ftw); (void)hipFree(c->d_cst); (void)hipFree(c->d_cstf);
  (void)hipFree(c->d_slot_map);
  drop_key_twins(c, nullptr);
  (void)hipFree(c->d_sk); (void)hipFree(c->d_pk); (void)hipFree(c->d_relin);
  for (auto &kv : c->d_galois) (void)hipFree(kv.second);
  (void)hipFree(c->ws);
  for (void *p : c->aux) (void)hipFree(p);
  // every block abc_hip_malloc ever handed out and that was not returned to the driver: cached ones and ones the caller
  // still holds (a caller that frees after destroying the context would otherwise leak them)
  for (auto &kv : c->block_size) (void)hipFree(kv.first);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->lane_fork) (void)hipEventDestroy(c->lane_fork);
  for (int i = 0; i < abc_hip_ctx::kMaxLanes; i++) {
    if (c->lane_join[i]) (void)hipEventDestroy(c->lane_join[i]);
    if (c->lane[i]) (void)hipStreamDestroy(c->lane[i]);
  }
  delete c;
}

int abc_hip_ctx_info(const abc_hip_ctx *c, int what) {
  if (!c) return -1;
  switch (what) {
    case 0: return c->scheme;
    case 1: return c->logn;
    case 2: return c->K;
    case 3: return c->L;
    case 4: return c->device;
    case 5: return c->nBsk;
    default: return -1;
  }
}

int abc_hip_set_stream(abc_hip_ctx *c, void *stream) {
  CTX_GUARD(c);
  NOT_CAPTURABLE(c, "abc_hip_set_stream");
  ABC_HIP_CHECK(hipStreamSynchronize(c->stream));
  c->stream = stream ? (hipStream_t)stream : c->own_stream;  // NULL selects the context's private stream again
  return 0;
}
int abc_hip_sync(abc_hip_ctx *c) {
  CTX_GUARD(c);
  NOT_CAPTURABLE(c, "abc_hip_sync");
  ABC_HIP_CHECK(hipStreamSynchronize(c->stream));
  return 0;
}

// Caching allocator: a freed buffer goes to a per-context free list (exact-size buckets) instead of back to the
// driver, and the next request of that size takes it -- no hipMalloc, no hipFree, no device synchronisation.  That is
// safe because every use of a context's buffers is ordered on the context's stream (the internal lanes fork from and
// join it inside each call, also on error paths: LaneScope in abc_kernels_fused.hip): whatever still runs on a recycled
// buffer was issued before its new owner's first use.  A buffer handed to ANOTHER context or stream is the caller's to
// order (header).  It matters to the plugin classes, where the interpreter clones / drops a ciphertext on every
// variable read.  The maps are guarded by a mutex (two host threads may share a context for allocation); the cache is
// flushed by abc_hip_trim, when the cap is reached, and whenever a hipMalloc of this context fails.
// (hipMallocAsync / hipFreeAsync were tried first and returned wrong results on some boxes of this pool when two
// contexts alternated -- analysis in DESIGN.md section 4b; ABC_HIP_SYNC_ALLOC=1 turns the cache off.)
static void *const kCapturing = (void *)(uintptr_t)1;
// pin bookkeeping (callers hold alloc_mu)
static void pin_add(abc_hip_ctx *c, void *p, void *owner) {
  auto &v = c->pin[p];
  for (void *o : v)
    if (o == owner) return;
  v.push_back(owner);
}
// drop `owner` from block p; a block nobody owns any more and that its user has already freed returns to the cache
static void pin_drop(abc_hip_ctx *c, void *p, void *owner) {
  auto it = c->pin.find(p);
  if (it == c->pin.end()) return;
  auto &v = it->second;
  for (size_t i = 0; i < v.size(); i++)
    if (v[i] == owner) {
      v[i] = v.back();
      v.pop_back();
      break;
    }
  if (!v.empty()) return;
  c->pin.erase(it);
  if (c->parked.erase(p)) {
    const size_t sz = c->block_size[p];
    c->free_blocks[sz].push_back(p);
    c->cached_bytes += sz;
  }
}

int abc_hip_malloc(abc_hip_ctx *c, void **d_ptr, size_t bytes) {
  CTX_GUARD(c);
  if (!bytes) bytes = 8;
  if (c->cache_alloc) {
    std::lock_guard<std::mutex> lock(c->alloc_mu);
    if (c->capture_active) {
      // inside a capture: a block born and freed in this capture first (safe: stream order inside the graph), then the
      // cache; never the driver (hipMalloc is not capturable) -- the sequence must have run once eagerly before
      auto cf = c->cap_free.find(bytes);
      if (cf != c->cap_free.end() && !cf->second.empty()) {
        *d_ptr = cf->second.back();
        cf->second.pop_back();
        return 0;
      }
    }
    auto it = c->free_blocks.find(bytes);
    if (it != c->free_blocks.end() && !it->second.empty()) {
      *d_ptr = it->second.back();
      it->second.pop_back();
      c->cached_bytes -= bytes;
      if (c->capture_active) {
        pin_add(c, *d_ptr, kCapturing);
        c->cap_born[*d_ptr] = true;
      }
      return 0;
    }
    if (c->capture_active) {
      set_error("allocation during graph capture found no cached buffer: run the sequence once eagerly first");
      return 1;
    }
  }
  ABC_HIP_CHECK(malloc_retry(c, d_ptr, bytes));
  if (c->cache_alloc) {
    std::lock_guard<std::mutex> lock(c->alloc_mu);
    c->block_size[*d_ptr] = bytes;
  }
  return 0;
}
int abc_hip_free(abc_hip_ctx *c, void *d_ptr) {
  CTX_GUARD(c);
  if (!d_ptr) return 0;
  if (c->cache_alloc) {
    bool over_cap = false;
    {
      std::lock_guard<std::mutex> lock(c->alloc_mu);
      auto it = c->block_size.find(d_ptr);
      if (it != c->block_size.end()) {
        if (c->capture_active) {
          if (c->cap_born.count(d_ptr)) {  // an intermediate of the circuit being recorded
            c->cap_free[it->second].push_back(d_ptr);
          } else {  // existed before: the graph reads it as an input on every replay -- pinned, never reused
            pin_add(c, d_ptr, kCapturing);
            c->parked[d_ptr] = true;
          }
          return 0;
        }
        if (c->pin.count(d_ptr)) {  // baked into a live graph: parked until abc_hip_graph_destroy
          c->parked[d_ptr] = true;
          return 0;
        }
        if (c->cached_bytes + it->second <= c->cache_cap) {
          c->free_blocks[it->second].push_back(d_ptr);
          c->cached_bytes += it->second;
          return 0;
        }
        c->block_size.erase(it);
        over_cap = true;
      }
    }
    // cap reached: exact-size buckets strand blocks when sizes vary, so give the whole cache back, not just this block
    if (over_cap && trim_cache(c)) return 1;
  }
  NOT_CAPTURABLE(c, "abc_hip_free of an uncached buffer");
  ABC_HIP_CHECK(hipStreamSynchronize(c->stream));
  ABC_HIP_CHECK(hipFree(d_ptr));
  return 0;
}
int abc_hip_trim(abc_hip_ctx *c) {
  CTX_GUARD(c);
  NOT_CAPTURABLE(c, "abc_hip_trim");
  return trim_cache(c);
}
size_t abc_hip_cached_bytes(abc_hip_ctx *c) {
  if (!c) return 0;
  std::lock_guard<std::mutex> lock(c->alloc_mu);
  return c->cached_bytes;
}
int abc_hip_ctx_reload_env(abc_hip_ctx *c) {
  CTX_GUARD(c);
  NOT_CAPTURABLE(c, "abc_hip_ctx_reload_env");
  ABC_HIP_CHECK(hipStreamSynchronize(c->stream));
  read_switches(c);
  return 0;
}
int abc_hip_memcpy_h2d(abc_hip_ctx *c, void *d, const void *h, size_t bytes) {
  CTX_GUARD(c);
  NOT_CAPTURABLE(c, "abc_hip_memcpy_h2d");
  ABC_HIP_CHECK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
  ABC_HIP_CHECK(hipStreamSynchronize(c->stream));
  return 0;
}
int abc_hip_memcpy_d2h(abc_hip_ctx *c, void *h, const void *d, size_t bytes) {
  CTX_GUARD(c);
  NOT_CAPTURABLE(c, "abc_hip_memcpy_d2h");
  ABC_HIP_CHECK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->stream));
  ABC_HIP_CHECK(hipStreamSynchronize(c->stream));
  return 0;
}
int abc_hip_memcpy_d2d(abc_hip_ctx *c, void *dst, const void *src, size_t bytes) {
  CTX_GUARD(c);
  ABC_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
  return 0;
}

// ---- keys ----
int abc_hip_keygen(abc_hip_ctx *c, uint64_t seed) {
  CTX_GUARD(c);
  NOT_CAPTURABLE(c, "abc_hip_keygen");
  try {
    return keygen(c, seed);
  } catch (const std::exception &e) {
    set_error(e.what());
    return 1;
  }
}
int abc_hip_keygen_secure(abc_hip_ctx *c) {
  CTX_GUARD(c);
  NOT_CAPTURABLE(c, "abc_hip_keygen_secure");
  try {
    return keygen_secure(c);
  } catch (const std::exception &e) {
    set_error(e.what());
    return 1;
  }
}
static int load_key(abc_hip_ctx *c, uint64_t **slot, const uint64_t *h, size_t words) {
  NOT_CAPTURABLE(c, "key upload");
  if (*slot) drop_key_twins(c, *slot);  // the mirror of the words about to be overwritten
  if (!*slot) ABC_HIP_CHECK(hipMalloc(slot, words * 8));
  ABC_HIP_CHECK(hipMemcpy(*slot, h, words * 8, hipMemcpyHostToDevice));
  return 0;
}
int abc_hip_load_secret_key(abc_hip_ctx *c, const uint64_t *h) { CTX_GUARD(c); return load_key(c, &c->d_sk, h, (size_t)c->K * c->n); }
int abc_hip_load_public_key(abc_hip_ctx *c, const uint64_t *h) { CTX_GUARD(c); return load_key(c, &c->d_pk, h, (size_t)2 * c->K * c->n); }
int abc_hip_load_relin_key(abc_hip_ctx *c, const uint64_t *h) { CTX_GUARD(c); return load_key(c, &c->d_relin, h, c->key_words()); }
int abc_hip_load_galois_key(abc_hip_ctx *c, uint32_t elt, const uint64_t *h) {
  CTX_GUARD(c);
  if (!(elt & 1) || elt >= 2u * (uint32_t)c->n) { set_error("Galois element must be odd and below 2N"); return 1; }
  uint64_t *slot = c->d_galois.count(elt) ? c->d_galois[elt] : nullptr;
  const bool fresh = (slot == nullptr);
  if (load_key(c, &slot, h, c->key_words())) return 1;
  c->d_galois[elt] = slot;
  if (fresh) c->galois_order.push_back(elt);
  return 0;
}
static int get_key(abc_hip_ctx *c, const uint64_t *d, uint64_t *h, size_t words, const char *what) {
  NOT_CAPTURABLE(c, "key download");
  if (!d) { set_error(std::string("key not present: ") + what); return 1; }
  ABC_HIP_CHECK(hipStreamSynchronize(c->stream));
  ABC_HIP_CHECK(hipMemcpy(h, d, words * 8, hipMemcpyDeviceToHost));
  return 0;
}
int abc_hip_get_secret_key(abc_hip_ctx *c, uint64_t *h) { CTX_GUARD(c); return get_key(c, c->d_sk, h, (size_t)c->K * c->n, "secret"); }
int abc_hip_get_public_key(abc_hip_ctx *c, uint64_t *h) { CTX_GUARD(c); return get_key(c, c->d_pk, h, (size_t)2 * c->K * c->n, "public"); }
int abc_hip_get_relin_key(abc_hip_ctx *c, uint64_t *h) { CTX_GUARD(c); return get_key(c, c->d_relin, h, c->key_words(), "relin"); }
int abc_hip_get_galois_key(abc_hip_ctx *c, uint32_t elt, uint64_t *h) {
  CTX_GUARD(c);
  auto it = c->d_galois.find(elt);
  return get_key(c, it == c->d_galois.end() ? nullptr : it->second, h, c->key_words(), "galois");
}
int abc_hip_num_galois_keys(abc_hip_ctx *c) { return c ? (int)c->galois_order.size() : 0; }
uint32_t abc_hip_galois_elt_at(abc_hip_ctx *c, int i) {
  if (!c || i < 0 || i >= (int)c->galois_order.size()) return 0;
  return c->galois_order[i];
}
uint32_t abc_hip_galois_elt_from_step(abc_hip_ctx *c, int step) { return c ? elt_from_step(c, step) : 0; }

// ---- encode / encrypt / decrypt ----
int abc_hip_batch_encode(abc_hip_ctx *c, const int64_t *v, uint64_t *p, size_t count) { CTX_GUARD(c); return batch_encode(c, v, p, count); }
int abc_hip_batch_decode(abc_hip_ctx *c, const uint64_t *p, int64_t *v, size_t count) { CTX_GUARD(c); return batch_decode(c, p, v, count); }
int abc_hip_encrypt(abc_hip_ctx *c, const uint64_t *p, uint64_t seed, uint64_t *ct, size_t count) { CTX_GUARD(c); NOT_CAPTURABLE(c, "abc_hip_encrypt"); return encrypt(c, p, seed, ct, count); }
int abc_hip_encrypt_secure(abc_hip_ctx *c, const uint64_t *p, uint64_t *ct, size_t count) { CTX_GUARD(c); NOT_CAPTURABLE(c, "abc_hip_encrypt_secure"); return encrypt_secure(c, p, ct, count); }
int abc_hip_decrypt(abc_hip_ctx *c, const uint64_t *ct, int size, int nl, uint64_t *p, size_t count) {
  CTX_GUARD(c);
  if (nl < 1 || nl > c->L) { set_error("decrypt: bad limb count"); return 1; }
  return decrypt(c, ct, size, nl, p, count);
}

// ---- evaluator ----
static int check_level(abc_hip_ctx *c, int nl) {
  if (nl < 1 || nl > c->L) { set_error("limb count out of range for this context"); return 1; }
  if (c->scheme == ABC_HIP_SCHEME_BFV && nl != c->L) { set_error("BFV ciphertexts live at the top level (nl = L)"); return 1; }
  return 0;
}
int abc_hip_add(abc_hip_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out, int size, int nl, size_t count) {
  CTX_GUARD(c);
  if (check_level(c, nl)) return 1;
  return launch_addsub(c, a, b, out, nl, count * size, 0);
}
int abc_hip_sub(abc_hip_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out, int size, int nl, size_t count) {
  CTX_GUARD(c);
  if (check_level(c, nl)) return 1;
  return launch_addsub(c, a, b, out, nl, count * size, 1);
}
int abc_hip_negate(abc_hip_ctx *c, const uint64_t *a, uint64_t *out, int size, int nl, size_t count) {
  CTX_GUARD(c);
  if (check_level(c, nl)) return 1;
  return launch_addsub(c, a, nullptr, out, nl, count * size, 2);
}
int abc_hip_multiply(abc_hip_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out3, int nl, size_t count) {
  CTX_GUARD(c);
  if (check_level(c, nl)) return 1;
  if (c->scheme == ABC_HIP_SCHEME_CKKS) return launch_ckks_tensor(c, a, b, out3, nl, count);
  return bfv_multiply(c, a, b, out3, count);
}
int abc_hip_relinearize(abc_hip_ctx *c, const uint64_t *ct3, uint64_t *out2, int nl, size_t count) {
  CTX_GUARD(c);
  if (check_level(c, nl)) return 1;
  return relinearize(c, ct3, out2, nl, count);
}
int abc_hip_mul_relin(abc_hip_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out, int nl, size_t count) {
  CTX_GUARD(c);
  if (check_level(c, nl)) return 1;
  if (!c->d_relin) { set_error("mul_relin: no relinearisation key"); return 1; }
  if (c->scheme == ABC_HIP_SCHEME_CKKS) {
    const int rc = ckks_mul_relin_fused(c, a, b, out, nl, count);
    if (rc >= 0) return rc;
  }
  if (c->scheme == ABC_HIP_SCHEME_BFV && bmul_applies(c)) return bmul_split(c, a, b, out, count, true);
  // generic path: size-3 product in arena 1, then key switch
  const size_t bytes = count * 3 * nl * (size_t)c->n * 8;
  if (ensure_aux(c, 1, bytes ? bytes : 8)) return 1;
  u64 *t3 = (u64 *)c->aux[1];
  int rc = (c->scheme == ABC_HIP_SCHEME_CKKS) ? launch_ckks_tensor(c, a, b, t3, nl, count) : bfv_multiply(c, a, b, t3, count);
  if (!rc) rc = relinearize(c, t3, out, nl, count);
  return rc;
}
int abc_hip_rotate(abc_hip_ctx *c, const uint64_t *in, uint64_t *out, int nl, int steps, size_t count) {
  CTX_GUARD(c);
  if (check_level(c, nl)) return 1;
  return rotate(c, in, out, nl, steps, count);
}
int abc_hip_apply_galois(abc_hip_ctx *c, const uint64_t *in, uint64_t *out, int nl, uint32_t elt, size_t count) {
  CTX_GUARD(c);
  if (check_level(c, nl)) return 1;
  if (in == out) { set_error("apply_galois: in-place not supported, use abc_hip_rotate"); return 1; }
  return apply_galois(c, in, out, nl, elt, count);
}
int abc_hip_multiply_plain(abc_hip_ctx *c, const uint64_t *ct, const uint64_t *plain, size_t plain_stride, uint64_t *out, int size,
                           int nl, size_t count) {
  CTX_GUARD(c);
  if (check_level(c, nl)) return 1;
  if (c->scheme == ABC_HIP_SCHEME_CKKS) return ckks_multiply_plain(c, ct, plain, plain_stride, out, size, nl, count);
  return bfv_multiply_plain(c, ct, plain, plain_stride, out, size, count);
}
int abc_hip_add_plain(abc_hip_ctx *c, const uint64_t *ct, const uint64_t *plain, size_t plain_stride, uint64_t *out, int size, int nl,
                      size_t count) {
  CTX_GUARD(c);
  if (check_level(c, nl)) return 1;
  if (c->scheme == ABC_HIP_SCHEME_CKKS) return ckks_add_plain(c, ct, plain, plain_stride, out, size, nl, count, 0);
  return bfv_addsub_plain(c, ct, plain, plain_stride, out, size, count, 0);
}
int abc_hip_sub_plain(abc_hip_ctx *c, const uint64_t *ct, const uint64_t *plain, size_t plain_stride, uint64_t *out, int size, int nl,
                      size_t count) {
  CTX_GUARD(c);
  if (check_level(c, nl)) return 1;
  if (c->scheme == ABC_HIP_SCHEME_CKKS) return ckks_add_plain(c, ct, plain, plain_stride, out, size, nl, count, 1);
  return bfv_addsub_plain(c, ct, plain, plain_stride, out, size, count, 1);
}
int abc_hip_rescale(abc_hip_ctx *c, const uint64_t *in, uint64_t *out, int size, int nl, size_t count) {
  CTX_GUARD(c);
  if (c->scheme != ABC_HIP_SCHEME_CKKS) { set_error("rescale is a CKKS operation"); return 1; }
  if (check_level(c, nl)) return 1;
  if (in == out) { set_error("rescale: d_out must not alias d_in (the output has one limb less per polynomial)"); return 1; }
  return launch_rescale(c, in, out, size, nl, count);
}
int abc_hip_mod_switch(abc_hip_ctx *c, const uint64_t *in, uint64_t *out, int size, int nl, size_t count) {
  CTX_GUARD(c);
  if (c->scheme != ABC_HIP_SCHEME_CKKS) { set_error("mod_switch is only implemented for CKKS"); return 1; }
  if (check_level(c, nl)) return 1;
  if (in == out) { set_error("mod_switch: d_out must not alias d_in (the output has one limb less per polynomial)"); return 1; }
  return launch_drop_last(c, in, out, size, nl, count);
}

// ---- raw pieces ----
static int ntt_entry(abc_hip_ctx *c, uint64_t *d, int kind, int index, size_t count, bool fwd) {
  int mid;
  if (kind == 0) {
    if (index < 0 || index >= c->K) { set_error("ntt: prime index out of range"); return 1; }
    mid = index;
  } else if (kind == 1) {
    if (c->scheme != ABC_HIP_SCHEME_BFV || index < 0 || index >= c->nBsk) { set_error("ntt: Bsk index out of range"); return 1; }
    mid = c->dc.id_bsk + index;
  } else if (kind == 2) {
    if (c->scheme != ABC_HIP_SCHEME_BFV) { set_error("ntt: no plaintext modulus in a CKKS context"); return 1; }
    mid = c->dc.id_t;
  } else {
    set_error("ntt: bad modulus kind");
    return 1;
  }
  LimbMap map{};
  map.id[0] = mid;
  return fwd ? launch_ntt_fwd(c, d, map, 1, count) : launch_ntt_inv(c, d, map, 1, count);
}
int abc_hip_ntt_forward(abc_hip_ctx *c, uint64_t *d, int kind, int index, size_t count) { CTX_GUARD(c); return ntt_entry(c, d, kind, index, count, true); }
int abc_hip_ntt_inverse(abc_hip_ctx *c, uint64_t *d, int kind, int index, size_t count) { CTX_GUARD(c); return ntt_entry(c, d, kind, index, count, false); }

int abc_hip_ntt_limbs(abc_hip_ctx *c, uint64_t *d, int nl, size_t polys, int inverse) {
  CTX_GUARD(c);
  if (nl < 1 || nl > c->L) { set_error("ntt_limbs: limb count out of range"); return 1; }
  const LimbMap map = key_limb_map(c, nl);
  return inverse ? launch_ntt_inv(c, d, map, nl, polys * nl) : launch_ntt_fwd(c, d, map, nl, polys * nl);
}

int abc_hip_keyswitch(abc_hip_ctx *c, const uint64_t *target, uint32_t key_kind, uint64_t *out2, int nl, size_t count) {
  CTX_GUARD(c);
  if (check_level(c, nl)) return 1;
  const uint64_t *key = nullptr;
  if (key_kind == 0) key = c->d_relin;
  else if (c->d_galois.count(key_kind)) key = c->d_galois[key_kind];
  if (!key) { set_error("keyswitch: key not present"); return 1; }
  return keyswitch(c, target, (size_t)nl * c->n, key, out2, nl, count, nullptr, 0, false);
}

// ---- graph capture ----
int abc_hip_graph_begin(abc_hip_ctx *c) {
  CTX_GUARD(c);
  ABC_HIP_CHECK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
  std::lock_guard<std::mutex> lock(c->alloc_mu);
  c->capture_active = true;
  return 0;
}
// hand every block the finished capture may have baked into the graph to `owner`; owner = nullptr (abandoned capture) releases
// what the capture had pinned
static void settle_capture(abc_hip_ctx *c, void *owner) {
  std::lock_guard<std::mutex> lock(c->alloc_mu);
  c->capture_active = false;
  for (auto &kv : c->cap_free)
    for (void *p : kv.second) c->parked[p] = true;  // born and freed inside the capture: nobody holds them any more
  c->cap_free.clear();
  c->cap_born.clear();
  std::vector<void *> mine;
  for (auto &kv : c->pin)
    for (void *o : kv.second)
      if (o == kCapturing) mine.push_back(kv.first);
  for (void *p : mine) {
    if (owner) pin_add(c, p, owner);
    pin_drop(c, p, kCapturing);
  }
  if (!owner) return;
  // every block still out with the caller: the recorded kernels may read it (an operand that existed before the capture and is
  // freed only later never passed through abc_hip_malloc / abc_hip_free while capture_active was set)
  std::unordered_map<void *, bool> cached;
  for (auto &kv : c->free_blocks)
    for (void *p : kv.second) cached[p] = true;
  for (auto &kv : c->block_size)
    if (!cached.count(kv.first)) pin_add(c, kv.first, owner);
}
int abc_hip_graph_end(abc_hip_ctx *c, void **out) {
  CTX_GUARD(c);
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture(c->stream, &graph);
  hipGraphExec_t exec = nullptr;
  if (e == hipSuccess) {
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    settle_capture(c, nullptr);
    set_error(std::string("graph capture failed: ") + hipGetErrorString(e));
    return 1;
  }
  settle_capture(c, (void *)exec);
  *out = exec;
  return 0;
}
int abc_hip_graph_launch(abc_hip_ctx *c, void *exec) {
  CTX_GUARD(c);
  ABC_HIP_CHECK(hipGraphLaunch((hipGraphExec_t)exec, c->stream));
  return 0;
}
int abc_hip_graph_destroy(abc_hip_ctx *c, void *exec) {
  CTX_GUARD(c);
  ABC_HIP_CHECK(hipStreamSynchronize(c->stream));
  ABC_HIP_CHECK(hipGraphExecDestroy((hipGraphExec_t)exec));
  std::lock_guard<std::mutex> lock(c->alloc_mu);
  std::vector<void *> mine;
  for (auto &kv : c->pin)
    for (void *o : kv.second)
      if (o == exec) mine.push_back(kv.first);
  for (void *p : mine) pin_drop(c, p, exec);  // unpin; what the caller had already freed goes back to the cache now
  return 0;
}

int abc_hip_microbench(abc_hip_ctx *c, int which, int iters, double *ms) { CTX_GUARD(c); return microbench(c, which, iters, ms); }

int abc_hip_timer_start(abc_hip_ctx *c) {
  CTX_GUARD(c);
  ABC_HIP_CHECK(hipEventRecord(c->ev0, c->stream));
  return 0;
}
int abc_hip_timer_stop(abc_hip_ctx *c, float *ms) {
  CTX_GUARD(c);
  ABC_HIP_CHECK(hipEventRecord(c->ev1, c->stream));
  ABC_HIP_CHECK(hipEventSynchronize(c->ev1));
  ABC_HIP_CHECK(hipEventElapsedTime(ms, c->ev0, c->ev1));
  return 0;
}

}  // extern "C"
