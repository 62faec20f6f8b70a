/*
 * oracle/orc_context.c -- TEST INFRASTRUCTURE (CPU oracle), see orc_internal.h header.
 *
 * Encryption-parameter context.  Restates what the reference sets up in
 * SealCiphertextFactory::setupSealContext (src/runtime/SealCiphertextFactory.cpp:72-100):
 *   scheme_type::bfv (:74), CoeffModulus::BFVDefault(N) (:80), PlainModulus::Batching(N, 20) (:83),
 *   SEALContext (:86)  -> modulus chain {data limbs | special prime}, NTT tables, BEHZ RNSTool.
 * [SEAL-recall: util/globals.cpp default_coeff_modulus_128, context.cpp, util/rns.cpp RNSTool::initialize]
 * CKKS (absent from the reference) follows SEAL's CoeffModulus::Create ordering.
 */
#include "orc_internal.h"
#include "oracle.h"

#include <stdlib.h>
#include <string.h>

/* [SEAL-recall: util/globals.cpp, 128-bit-security defaults] */
int orc_default_bfv_primes(size_t n, uint64_t *out) {
  static const uint64_t p1024[] = {0x7e00001ull};
  static const uint64_t p2048[] = {0x3fffffff000001ull};
  static const uint64_t p4096[] = {0xffffee001ull, 0xffffc4001ull, 0x1ffffe0001ull};
  static const uint64_t p8192[] = {0x7fffffd8001ull, 0x7fffffc8001ull, 0xfffffffc001ull, 0xffffff6c001ull,
                                   0xfffffebc001ull};
  static const uint64_t p16384[] = {0xfffffffd8001ull,  0xfffffffa0001ull,  0xfffffff00001ull,
                                    0x1fffffff68001ull, 0x1fffffff50001ull, 0x1ffffffee8001ull,
                                    0x1ffffffea0001ull, 0x1ffffffe88001ull, 0x1ffffffe48001ull};
  static const uint64_t p32768[] = {0x7fffffffe90001ull, 0x7fffffffbf0001ull, 0x7fffffffbd0001ull,
                                    0x7fffffffba0001ull, 0x7fffffffaa0001ull, 0x7fffffffa50001ull,
                                    0x7fffffff9f0001ull, 0x7fffffff7e0001ull, 0x7fffffff770001ull,
                                    0x7fffffff380001ull, 0x7fffffff330001ull, 0x7fffffff2d0001ull,
                                    0x7fffffff170001ull, 0x7fffffff150001ull, 0x7ffffffef00001ull,
                                    0xfffffffff70001ull};
  const uint64_t *src; int cnt;
  switch (n) {
    case 1024: src = p1024; cnt = 1; break;
    case 2048: src = p2048; cnt = 1; break;
    case 4096: src = p4096; cnt = 3; break;
    case 8192: src = p8192; cnt = 5; break;
    case 16384: src = p16384; cnt = 9; break;
    case 32768: src = p32768; cnt = 16; break;
    default: return -1;
  }
  for (int i = 0; i < cnt; i++) out[i] = src[i];
  return cnt;
}

/* [SEAL-recall: CoeffModulus::Create(N, bit_sizes)] for each distinct bit size the primes are
 * generated in decreasing order and handed out from the BACK of that list. */
int orc_create_primes(size_t n, const int *bit_sizes, int count, uint64_t *out) {
  for (int i = 0; i < count; i++) out[i] = 0;
  for (int i = 0; i < count; i++) {
    if (out[i]) continue;
    int bits = bit_sizes[i], same = 0;
    for (int k = 0; k < count; k++) if (bit_sizes[k] == bits) same++;
    uint64_t tmp[ORC_MAX_LIMBS];
    if (orc_get_primes(n, bits, same, tmp)) return -1;
    int back = same - 1;
    for (int k = 0; k < count; k++) if (bit_sizes[k] == bits) out[k] = tmp[back--];
  }
  return 0;
}

/* [SEAL-recall: PlainModulus::Batching(N, bits)] = largest bits-bit prime = 1 mod 2N */
uint64_t orc_plain_modulus_batching(size_t n, int bits) {
  uint64_t p = 0;
  if (orc_get_primes(n, bits, 1, &p)) return 0;
  return p;
}

/* bit length of the product of moduli (tiny bignum) */
static int product_bit_count(const orc_mod *m, int cnt) {
  uint64_t w[ORC_MAX_LIMBS + 1];
  memset(w, 0, sizeof(w));
  w[0] = 1;
  int len = 1;
  for (int i = 0; i < cnt; i++) {
    uint64_t carry = 0;
    for (int k = 0; k < len; k++) {
      u128 p = (u128)w[k] * m[i].q + carry;
      w[k] = (uint64_t)p;
      carry = (uint64_t)(p >> 64);
    }
    if (carry) w[len++] = carry;
  }
  int top = len - 1;
  int bits = 0;
  uint64_t v = w[top];
  while (v) { bits++; v >>= 1; }
  return top * 64 + bits;
}

static uint64_t prod_mod(const orc_mod *m, int cnt, uint64_t p) {
  uint64_t v = 1 % p;
  for (int i = 0; i < cnt; i++) v = orc_mul_mod(v, m[i].q % p, p);
  return v;
}

/* [SEAL-recall: RNSTool::initialize] */
static orc_behz *behz_create(int logn, const orc_mod *q, int nq, uint64_t t) {
  orc_behz *b = (orc_behz *)calloc(1, sizeof(orc_behz));
  size_t n = (size_t)1 << logn;
  b->nq = nq;
  for (int i = 0; i < nq; i++) b->q[i] = q[i];
  orc_mod_init(&b->t, t);
  int t_bits = 0; { uint64_t v = t; while (v) { t_bits++; v >>= 1; } }
  int total_bits = product_bit_count(q, nq);
  /* SEAL takes m_sk, gamma and B from the 61-bit NTT primes.  ORC_BEHZ_AUX_BITS=<bits> (tests only) builds m_sk and B
   * from primes of another size instead -- same sizing rule, primes of the ciphertext modulus skipped, gamma unchanged
   * (it only serves decryption) -- to demonstrate that the q-residues a BFV multiply returns do not depend on which
   * auxiliary primes carry the intermediate values (tests/test_oracle_internal.py). */
  int aux_bits = 61;
  { const char *e = getenv("ORC_BEHZ_AUX_BITS"); if (e && atoi(e) >= 30 && atoi(e) <= 61) aux_bits = atoi(e); }
  int nB = nq;
  if (32 + t_bits + total_bits >= aux_bits * nq + aux_bits) nB++;
  b->nB = nB; b->nBsk = nB + 1;
  uint64_t aux[ORC_MAX_LIMBS + 2];
  uint64_t seal_aux[2];
  if (orc_get_primes(n, 61, 2, seal_aux)) { free(b); return NULL; }
  if (aux_bits == 61) {
    if (orc_get_primes(n, 61, nB + 2, aux)) { free(b); return NULL; }
  } else {
    uint64_t cand[2 * ORC_MAX_LIMBS + 4];
    if (orc_get_primes(n, aux_bits, nB + 1 + nq + 1, cand)) { free(b); return NULL; }
    int got = 0;
    aux[1] = seal_aux[1];
    for (int i = 0; i < nB + 1 + nq + 1 && got < nB + 1; i++) {
      int clash = 0;
      for (int k = 0; k < nq; k++) clash |= (cand[i] == q[k].q);
      if (clash) continue;
      aux[got == 0 ? 0 : got + 1] = cand[i];
      got++;
    }
    if (got < nB + 1) { free(b); return NULL; }
  }
  orc_mod_init(&b->m_sk, aux[0]);
  orc_mod_init(&b->gamma, aux[1]);
  for (int i = 0; i < nB; i++) { orc_mod_init(&b->B[i], aux[2 + i]); b->Bsk[i] = b->B[i]; }
  b->Bsk[nB] = b->m_sk;
  orc_mod_init(&b->m_tilde, (uint64_t)1 << 32);

  b->Bsk_ntt = (orc_ntt *)calloc(b->nBsk, sizeof(orc_ntt));
  for (int j = 0; j < b->nBsk; j++) orc_ntt_init(&b->Bsk_ntt[j], logn, b->Bsk[j].q);

  orc_bconv_init(&b->q_to_Bsk, b->q, nq, b->Bsk, b->nBsk);
  orc_bconv_init(&b->q_to_mtilde, b->q, nq, &b->m_tilde, 1);
  orc_bconv_init(&b->B_to_q, b->B, nB, b->q, nq);
  orc_bconv_init(&b->B_to_msk, b->B, nB, &b->m_sk, 1);
  orc_mod tg[2] = {b->t, b->gamma};
  orc_bconv_init(&b->q_to_tgamma, b->q, nq, tg, 2);

  uint64_t mt = b->m_tilde.q;
  for (int i = 0; i < nq; i++) {
    b->mtilde_mod_q[i] = mt % q[i].q;
    b->B_mod_q[i] = prod_mod(b->B, nB, q[i].q);
    b->tgamma_mod_q[i] = orc_mul_mod(t % q[i].q, b->gamma.q % q[i].q, q[i].q);
  }
  uint64_t q_mod_mt = prod_mod(q, nq, mt);
  b->neg_inv_q_mod_mtilde = (mt - orc_inv_mod(q_mod_mt, mt)) % mt;
  for (int j = 0; j < b->nBsk; j++) {
    uint64_t pj = b->Bsk[j].q;
    b->q_mod_Bsk[j] = prod_mod(q, nq, pj);
    b->inv_q_mod_Bsk[j] = orc_inv_mod(b->q_mod_Bsk[j], pj);
    b->inv_mtilde_mod_Bsk[j] = orc_inv_mod(mt % pj, pj);
  }
  b->inv_B_mod_msk = orc_inv_mod(prod_mod(b->B, nB, b->m_sk.q), b->m_sk.q);
  b->neg_inv_q_mod_t = orc_neg_mod(orc_inv_mod(prod_mod(q, nq, t), t), t);
  b->neg_inv_q_mod_gamma = orc_neg_mod(orc_inv_mod(prod_mod(q, nq, b->gamma.q), b->gamma.q), b->gamma.q);
  b->inv_gamma_mod_t = orc_inv_mod(b->gamma.q % t, t);
  return b;
}

static void behz_free(orc_behz *b) {
  if (!b) return;
  for (int j = 0; j < b->nBsk; j++) orc_ntt_free(&b->Bsk_ntt[j]);
  free(b->Bsk_ntt);
  free(b);
}

orc_ctx *orc_ctx_create(int scheme, int logn, const uint64_t *primes, int nprimes, uint64_t plain_modulus) {
  if (nprimes < 2 || nprimes > ORC_MAX_LIMBS) return NULL;
  orc_ctx *c = (orc_ctx *)calloc(1, sizeof(orc_ctx));
  c->scheme = scheme; c->logn = logn; c->n = (size_t)1 << logn;
  c->nkey = nprimes; c->L = nprimes - 1;
  c->ntt = (orc_ntt *)calloc(nprimes, sizeof(orc_ntt));
  for (int i = 0; i < nprimes; i++) {
    orc_mod_init(&c->qmod[i], primes[i]);
    if (orc_ntt_init(&c->ntt[i], logn, primes[i])) { orc_ctx_destroy(c); return NULL; }
  }
  uint64_t qsp = primes[nprimes - 1];
  for (int i = 0; i < c->L; i++) c->inv_special_mod_q[i] = orc_inv_mod(qsp % primes[i], primes[i]);
  for (int l = 1; l < c->L; l++)
    for (int i = 0; i < l; i++) c->inv_qlast_mod_q[l][i] = orc_inv_mod(primes[l] % primes[i], primes[i]);
  if (scheme == ORC_SCHEME_BFV) {
    uint64_t t = plain_modulus;
    orc_mod_init(&c->t, t);
    c->t_ntt = (orc_ntt *)calloc(1, sizeof(orc_ntt));
    if (orc_ntt_init(c->t_ntt, logn, t)) { orc_ctx_destroy(c); return NULL; }
    /* [SEAL-recall: BatchEncoder::populate_matrix_reps_index_map] */
    size_t n = c->n, row = n >> 1, m = n << 1;
    c->slot_map = (size_t *)malloc(n * sizeof(size_t));
    uint64_t pos = 1;
    for (size_t i = 0; i < row; i++) {
      uint64_t i1 = (pos - 1) >> 1, i2 = (m - pos - 1) >> 1;
      c->slot_map[i] = orc_bitrev((uint32_t)i1, logn);
      c->slot_map[row | i] = orc_bitrev((uint32_t)i2, logn);
      pos = (pos * 3) & (m - 1);
    }
    /* [SEAL-recall: context.cpp] coeff_div_plain_modulus = floor(q/t), q mod t, thresholds */
    c->q_mod_t = prod_mod(c->qmod, c->L, t);
    c->upper_half_threshold = (t + 1) >> 1;
    for (int i = 0; i < c->L; i++) {
      uint64_t qi = primes[i];
      /* floor(q/t) = (q - (q mod t)) / t  =>  mod q_i: (-(q mod t)) * t^-1 */
      uint64_t r = c->q_mod_t % qi;
      c->coeff_div_plain[i] = orc_mul_mod(orc_neg_mod(r, qi), orc_inv_mod(t % qi, qi), qi);
      c->upper_half_increment[i] = qi - t;
    }
    c->behz = behz_create(logn, c->qmod, c->L, t);
    if (!c->behz) { orc_ctx_destroy(c); return NULL; }
  }
  return c;
}

void orc_ctx_destroy(orc_ctx *c) {
  if (!c) return;
  if (c->ntt) { for (int i = 0; i < c->nkey; i++) orc_ntt_free(&c->ntt[i]); free(c->ntt); }
  if (c->t_ntt) { orc_ntt_free(c->t_ntt); free(c->t_ntt); }
  free(c->slot_map);
  behz_free(c->behz);
  free(c->sk_ntt); free(c->pk); free(c->relin);
  for (int i = 0; i < c->ngal; i++) free(c->gal_key[i]);
  free(c);
}

int orc_ctx_info(const orc_ctx *c, int what) {
  switch (what) {
    case 0: return c->scheme;
    case 1: return c->logn;
    case 2: return c->nkey;
    case 3: return c->L;
    case 4: return c->behz ? c->behz->nBsk : 0;
    case 5: return c->ngal;
    default: return -1;
  }
}
uint64_t orc_ctx_prime(const orc_ctx *c, int i) { return c->qmod[i].q; }
uint64_t orc_ctx_plain_modulus(const orc_ctx *c) { return c->t.q; }
uint64_t orc_ctx_ntt_root(const orc_ctx *c, int i) { return c->ntt[i].root; }
uint64_t orc_ctx_behz_prime(const orc_ctx *c, int which) {
  /* 0: m_sk, 1: gamma, 2+i: B_i */
  if (!c->behz) return 0;
  if (which == 0) return c->behz->m_sk.q;
  if (which == 1) return c->behz->gamma.q;
  return c->behz->B[which - 2].q;
}
