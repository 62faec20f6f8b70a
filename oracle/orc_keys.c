/*
 * oracle/orc_keys.c -- TEST INFRASTRUCTURE (CPU oracle), see orc_internal.h header.
 *
 * Key generation, BatchEncoder, encryption and decryption.  Restates what the reference calls at
 *   SealCiphertextFactory.cpp:89-93  KeyGenerator: secret_key / create_public_key /
 *                                    create_galois_keys (all default elements) / create_relin_keys
 *   SealCiphertextFactory.cpp:130    BatchEncoder::encode    (after expandVector pad-with-last, :102-115)
 *   SealCiphertextFactory.cpp:12     Encryptor::encrypt      (public-key)
 *   SealCiphertextFactory.cpp:150-151 Decryptor::decrypt + BatchEncoder::decode
 *   SealCiphertext.cpp:80-83         Decryptor::invariant_noise_budget
 * [SEAL-recall: keygenerator.cpp, util/rlwe.cpp, encryptor.cpp, decryptor.cpp, batchencoder.cpp,
 *  util/rns.cpp (divide_and_round_q_last_inplace, decrypt_scale_and_round), util/scalingvariant.cpp]
 *
 * Randomness: SEAL draws from a Blake2/Shake PRNG with a random seed, so key and ciphertext
 * bits are never reproducible against SEAL; this repo fixes its own sampling spec
 * (DESIGN.md "Sampling spec") so that oracle and HIP product agree bit-for-bit given a seed.
 */
#include "orc_internal.h"
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------- sampling spec ---------- */
static int sample_ternary(orc_rng *r) { /* uniform in {-1,0,1} */
  for (;;) {
    uint64_t x = orc_rng_next(r);
    if (x == ~(uint64_t)0) continue; /* 2^64 mod 3 == 1: reject the single excess value */
    return (int)(x % 3) - 1;
  }
}
static int sample_cbd(orc_rng *r) { /* centred binomial, 21 vs 21 bits [SEAL-recall: sample_poly_cbd] */
  uint64_t x = orc_rng_next(r);
  return __builtin_popcountll(x & 0x1FFFFF) - __builtin_popcountll((x >> 21) & 0x1FFFFF);
}
static uint64_t sample_uniform(orc_rng *r, uint64_t q) { /* [SEAL-recall: sample_poly_uniform] */
  uint64_t max_multiple = ~(uint64_t)0 - (~(uint64_t)0 % q) - 1;
  uint64_t x;
  do { x = orc_rng_next(r); } while (x >= max_multiple);
  return x % q;
}
static void small_to_rns(const int *small, size_t n, uint64_t q, uint64_t *out) {
  for (size_t k = 0; k < n; k++) out[k] = small[k] < 0 ? q - (uint64_t)(-small[k]) : (uint64_t)small[k];
}

static void dyadic_mul(const uint64_t *a, const uint64_t *b, uint64_t *o, size_t n, const orc_mod *m) {
  for (size_t k = 0; k < n; k++) o[k] = orc_mulmod_b(a[k], b[k], m);
}

/* [SEAL-recall: GaloisTool::generate_table_ntt / apply_galois_ntt] */
static void galois_ntt_one(const uint64_t *in, uint64_t *out, int logn, uint32_t elt) {
  size_t n = (size_t)1 << logn;
  for (size_t i = 0; i < n; i++) {
    uint32_t rev = orc_bitrev((uint32_t)(i + n), logn + 1);
    uint64_t idx = (((uint64_t)elt * rev) >> 1) & (n - 1);
    out[i] = in[orc_bitrev((uint32_t)idx, logn)];
  }
}

/* [SEAL-recall: util::encrypt_zero_symmetric, NTT form, key level]  (c0,c1) = (-(a*s+e), a) */
static void encrypt_zero_symmetric(const orc_ctx *c, orc_rng *rng, uint64_t *ct /*[2][nkey][n]*/) {
  size_t n = c->n;
  int K = c->nkey;
  uint64_t *c0 = ct, *c1 = ct + (size_t)K * n;
  for (int j = 0; j < K; j++)
    for (size_t k = 0; k < n; k++) c1[(size_t)j * n + k] = sample_uniform(rng, c->qmod[j].q);
  int *e = (int *)malloc(n * sizeof(int));
  for (size_t k = 0; k < n; k++) e[k] = sample_cbd(rng);
  uint64_t *tmp = (uint64_t *)malloc(n * sizeof(uint64_t));
  for (int j = 0; j < K; j++) {
    uint64_t q = c->qmod[j].q;
    small_to_rns(e, n, q, tmp);
    orc_ntt_fwd(tmp, &c->ntt[j]);
    const uint64_t *s = c->sk_ntt + (size_t)j * n;
    for (size_t k = 0; k < n; k++) {
      uint64_t as = orc_mulmod_b(c1[(size_t)j * n + k], s[k], &c->qmod[j]);
      c0[(size_t)j * n + k] = orc_neg_mod(orc_add_mod(as, tmp[k], q), q);
    }
  }
  free(tmp); free(e);
}

/* [SEAL-recall: KeyGenerator::generate_one_kswitch_key] new_key: [nkey][n] NTT form */
static void make_kswitch_key(const orc_ctx *c, orc_rng *rng, const uint64_t *new_key, uint64_t *out) {
  size_t n = c->n;
  int K = c->nkey, L = c->L;
  uint64_t qsp = c->qmod[K - 1].q;
  for (int i = 0; i < L; i++) {
    uint64_t *ki = out + (size_t)i * 2 * K * n;
    encrypt_zero_symmetric(c, rng, ki);
    uint64_t qi = c->qmod[i].q, factor = qsp % qi;
    uint64_t *dst = ki + (size_t)i * n; /* component 0, limb i */
    const uint64_t *nk = new_key + (size_t)i * n;
    for (size_t k = 0; k < n; k++) dst[k] = orc_add_mod(dst[k], orc_mulmod_b(nk[k], factor, &c->qmod[i]), qi);
  }
}

uint32_t orc_galois_elt_from_step(const orc_ctx *c, int step) {
  /* [SEAL-recall: GaloisTool::get_elt_from_step] */
  uint32_t n = (uint32_t)c->n;
  uint64_t m = 2ull * n;
  if (step == 0) return (uint32_t)(m - 1);
  int neg = step < 0;
  uint32_t pos = (uint32_t)(neg ? -step : step);
  if (pos >= (n >> 1)) return 0; /* "step count too large" */
  int s = neg ? (int)(n >> 1) - (int)pos : (int)pos;
  uint64_t g = 1;
  for (int i = 0; i < s; i++) g = (g * 3) & (m - 1);
  return (uint32_t)g;
}

int orc_naf(int value, int *out) {
  /* [SEAL-recall: util::naf] */
  int cnt = 0, sign = value < 0;
  value = abs(value);
  for (int i = 0; value; i++) {
    int zi = (value & 1) ? 2 - (value & 3) : 0;
    value = (value - zi) >> 1;
    if (zi) out[cnt++] = (sign ? -zi : zi) * (1 << i);
  }
  return cnt;
}

int orc_keygen(orc_ctx *c, uint64_t seed) {
  size_t n = c->n;
  int K = c->nkey, L = c->L;
  orc_rng rng;
  orc_rng_seed(&rng, seed);
  free(c->sk_ntt); free(c->pk); free(c->relin);
  for (int i = 0; i < c->ngal; i++) free(c->gal_key[i]);
  c->ngal = 0;
  /* secret key: ternary, stored in NTT form at key level [SEAL-recall: KeyGenerator::generate_sk] */
  int *s = (int *)malloc(n * sizeof(int));
  for (size_t k = 0; k < n; k++) s[k] = sample_ternary(&rng);
  c->sk_ntt = (uint64_t *)malloc((size_t)K * n * 8);
  for (int j = 0; j < K; j++) {
    small_to_rns(s, n, c->qmod[j].q, c->sk_ntt + (size_t)j * n);
    orc_ntt_fwd(c->sk_ntt + (size_t)j * n, &c->ntt[j]);
  }
  free(s);
  /* public key */
  c->pk = (uint64_t *)malloc((size_t)2 * K * n * 8);
  encrypt_zero_symmetric(c, &rng, c->pk);
  /* relin key: key-switch key for s^2 [SEAL-recall: KeyGenerator::create_relin_keys(count=1)] */
  uint64_t *s2 = (uint64_t *)malloc((size_t)K * n * 8);
  for (int j = 0; j < K; j++)
    dyadic_mul(c->sk_ntt + (size_t)j * n, c->sk_ntt + (size_t)j * n, s2 + (size_t)j * n, n, &c->qmod[j]);
  size_t key_words = (size_t)L * 2 * K * n;
  c->relin = (uint64_t *)malloc(key_words * 8);
  make_kswitch_key(c, &rng, s2, c->relin);
  /* Galois keys for all default elements [SEAL-recall: GaloisTool::get_elts_all] */
  uint64_t m = 2ull * n;
  uint32_t elts[ORC_MAX_GALOIS];
  int ne = 0;
  elts[ne++] = (uint32_t)(m - 1);
  uint64_t pos = 3, neg = orc_inv_mod(3, m);
  for (int i = 0; i < c->logn - 1; i++) {
    elts[ne++] = (uint32_t)pos; pos = (pos * pos) & (m - 1);
    elts[ne++] = (uint32_t)neg; neg = (neg * neg) & (m - 1);
  }
  int nk = 0;
  for (int e = 0; e < ne; e++) {
    /* [SEAL-recall: KeyGenerator::create_galois_keys] "do we already have the key?" -> skip
     * (3^(N/4) == 3^-(N/4) mod 2N, so the default list names that element twice) */
    int dup = 0;
    for (int d = 0; d < nk; d++) dup |= (c->gal_elt[d] == elts[e]);
    if (dup) continue;
    for (int j = 0; j < K; j++) galois_ntt_one(c->sk_ntt + (size_t)j * n, s2 + (size_t)j * n, c->logn, elts[e]);
    c->gal_elt[nk] = elts[e];
    c->gal_key[nk] = (uint64_t *)malloc(key_words * 8);
    make_kswitch_key(c, &rng, s2, c->gal_key[nk]);
    nk++;
  }
  c->ngal = nk;
  free(s2);
  return 0;
}

int orc_get_secret_key(const orc_ctx *c, uint64_t *out) {
  if (!c->sk_ntt) return -1;
  memcpy(out, c->sk_ntt, (size_t)c->nkey * c->n * 8);
  return 0;
}
int orc_get_public_key(const orc_ctx *c, uint64_t *out) {
  if (!c->pk) return -1;
  memcpy(out, c->pk, (size_t)2 * c->nkey * c->n * 8);
  return 0;
}
int orc_get_relin_key(const orc_ctx *c, uint64_t *out) {
  if (!c->relin) return -1;
  memcpy(out, c->relin, (size_t)c->L * 2 * c->nkey * c->n * 8);
  return 0;
}
int orc_num_galois(const orc_ctx *c) { return c->ngal; }
uint32_t orc_galois_elt_at(const orc_ctx *c, int i) { return c->gal_elt[i]; }
int orc_get_galois_key(const orc_ctx *c, uint32_t elt, uint64_t *out) {
  for (int i = 0; i < c->ngal; i++)
    if (c->gal_elt[i] == elt) {
      memcpy(out, c->gal_key[i], (size_t)c->L * 2 * c->nkey * c->n * 8);
      return 0;
    }
  return -1;
}

/* ---------- BatchEncoder ---------- */
int orc_batch_encode(const orc_ctx *c, const int64_t *values, size_t count, uint64_t *plain) {
  /* [SEAL-recall: BatchEncoder::encode(vector<int64_t>)] */
  if (c->scheme != ORC_SCHEME_BFV || count > c->n) return -1;
  uint64_t t = c->t.q;
  for (size_t i = 0; i < count; i++) {
    int64_t v = values[i];
    plain[c->slot_map[i]] = v < 0 ? t + (uint64_t)v : (uint64_t)v;
  }
  for (size_t i = count; i < c->n; i++) plain[c->slot_map[i]] = 0;
  orc_ntt_inv(plain, c->t_ntt);
  return 0;
}
int orc_batch_decode(const orc_ctx *c, const uint64_t *plain, int64_t *values) {
  /* [SEAL-recall: BatchEncoder::decode(vector<int64_t>)] */
  if (c->scheme != ORC_SCHEME_BFV) return -1;
  size_t n = c->n;
  uint64_t t = c->t.q, half = t >> 1;
  uint64_t *tmp = (uint64_t *)malloc(n * 8);
  memcpy(tmp, plain, n * 8);
  orc_ntt_fwd(tmp, c->t_ntt);
  for (size_t i = 0; i < n; i++) {
    uint64_t v = tmp[c->slot_map[i]];
    values[i] = v > half ? (int64_t)v - (int64_t)t : (int64_t)v;
  }
  free(tmp);
  return 0;
}

/* ---------- encryption ---------- */
/* [SEAL-recall: util::encrypt_zero_asymmetric at key level + RNSTool::divide_and_round_q_last(_ntt)_inplace]
 * out: [2][L][n] at the top data level, coefficient form (BFV) or NTT form (CKKS). */
static void encrypt_zero_asymmetric_modswitch(const orc_ctx *c, orc_rng *rng, uint64_t *out) {
  size_t n = c->n;
  int K = c->nkey, L = c->L;
  int ntt_form = (c->scheme == ORC_SCHEME_CKKS);
  int *u = (int *)malloc(n * sizeof(int));
  int *e = (int *)malloc(n * sizeof(int));
  for (size_t k = 0; k < n; k++) u[k] = sample_ternary(rng);
  uint64_t *un = (uint64_t *)malloc((size_t)K * n * 8);
  for (int j = 0; j < K; j++) {
    small_to_rns(u, n, c->qmod[j].q, un + (size_t)j * n);
    orc_ntt_fwd(un + (size_t)j * n, &c->ntt[j]);
  }
  uint64_t *tmp = (uint64_t *)malloc((size_t)K * n * 8);
  uint64_t *en = (uint64_t *)malloc(n * 8);
  uint64_t qk = c->qmod[K - 1].q, half = qk >> 1;
  for (int p = 0; p < 2; p++) {
    for (size_t k = 0; k < n; k++) e[k] = sample_cbd(rng);
    for (int j = 0; j < K; j++) {
      uint64_t *tj = tmp + (size_t)j * n;
      uint64_t q = c->qmod[j].q;
      dyadic_mul(un + (size_t)j * n, c->pk + ((size_t)p * K + j) * n, tj, n, &c->qmod[j]);
      small_to_rns(e, n, q, en);
      if (ntt_form) {
        orc_ntt_fwd(en, &c->ntt[j]);
      } else {
        orc_ntt_inv(tj, &c->ntt[j]);
      }
      for (size_t k = 0; k < n; k++) tj[k] = orc_add_mod(tj[k], en[k], q);
    }
    /* drop the special prime with rounding */
    uint64_t *last = tmp + (size_t)(K - 1) * n;
    if (ntt_form) orc_ntt_inv(last, &c->ntt[K - 1]);
    for (size_t k = 0; k < n; k++) last[k] = orc_add_mod(last[k], half % qk, qk);
    for (int j = 0; j < L; j++) {
      uint64_t q = c->qmod[j].q;
      uint64_t half_mod = half % q;
      for (size_t k = 0; k < n; k++) en[k] = orc_sub_mod(last[k] % q, half_mod, q);
      if (ntt_form) orc_ntt_fwd(en, &c->ntt[j]);
      uint64_t *tj = tmp + (size_t)j * n;
      uint64_t *dst = out + ((size_t)p * L + j) * n;
      for (size_t k = 0; k < n; k++)
        dst[k] = orc_mulmod_b(orc_sub_mod(tj[k], en[k], q), c->inv_special_mod_q[j], &c->qmod[j]);
    }
  }
  free(en); free(tmp); free(un); free(e); free(u);
}

/* [SEAL-recall: util::multiply_add_plain_with_scaling_variant / multiply_sub_...]
 * adds (sub=0) or subtracts (sub=1) round(q*m/t) to poly [L][n] */
static void scaled_plain_addsub(const orc_ctx *c, const uint64_t *plain, uint64_t *poly, int sub) {
  size_t n = c->n;
  uint64_t t = c->t.q;
  for (size_t k = 0; k < n; k++) {
    u128 numer = (u128)plain[k] * c->q_mod_t + c->upper_half_threshold;
    uint64_t fix = (uint64_t)(numer / t);
    for (int j = 0; j < c->L; j++) {
      uint64_t q = c->qmod[j].q;
      uint64_t scaled = orc_add_mod(orc_mulmod_b(plain[k], c->coeff_div_plain[j], &c->qmod[j]), fix % q, q);
      uint64_t *p = poly + (size_t)j * n + k;
      *p = sub ? orc_sub_mod(*p, scaled, q) : orc_add_mod(*p, scaled, q);
    }
  }
}
void orc_scaled_plain_addsub(const orc_ctx *c, const uint64_t *plain, uint64_t *poly, int sub) {
  scaled_plain_addsub(c, plain, poly, sub);
}

int orc_bfv_encrypt(const orc_ctx *c, const uint64_t *plain, uint64_t seed, uint64_t *ct) {
  if (c->scheme != ORC_SCHEME_BFV || !c->pk) return -1;
  orc_rng rng;
  orc_rng_seed(&rng, seed);
  encrypt_zero_asymmetric_modswitch(c, &rng, ct);
  scaled_plain_addsub(c, plain, ct, 0);
  return 0;
}

int orc_ckks_encrypt(const orc_ctx *c, const uint64_t *plain, uint64_t seed, uint64_t *ct) {
  if (c->scheme != ORC_SCHEME_CKKS || !c->pk) return -1;
  orc_rng rng;
  orc_rng_seed(&rng, seed);
  encrypt_zero_asymmetric_modswitch(c, &rng, ct);
  for (int j = 0; j < c->L; j++)
    for (size_t k = 0; k < c->n; k++) {
      size_t o = (size_t)j * c->n + k;
      ct[o] = orc_add_mod(ct[o], plain[o], c->qmod[j].q);
    }
  return 0;
}

/* ---------- decryption ---------- */
/* phase = c0 + c1*s + c2*s^2 ... mod q_j for j<nl; result in the ciphertext's own form */
static void dot_ct_sk(const orc_ctx *c, const uint64_t *ct, int size, int nl, int ntt_form, uint64_t *phase) {
  size_t n = c->n;
  uint64_t *tmp = (uint64_t *)malloc(n * 8);
  uint64_t *spow = (uint64_t *)malloc(n * 8);
  for (int j = 0; j < nl; j++) {
    const orc_mod *m = &c->qmod[j];
    uint64_t *ph = phase + (size_t)j * n;
    const uint64_t *s = c->sk_ntt + (size_t)j * n;
    memset(ph, 0, n * 8);
    memcpy(spow, s, n * 8);
    for (int p = 1; p < size; p++) {
      memcpy(tmp, ct + ((size_t)p * nl + j) * n, n * 8);
      if (!ntt_form) orc_ntt_fwd(tmp, &c->ntt[j]);
      for (size_t k = 0; k < n; k++) ph[k] = orc_add_mod(ph[k], orc_mulmod_b(tmp[k], spow[k], m), m->q);
      if (p + 1 < size) dyadic_mul(spow, s, spow, n, m);
    }
    if (!ntt_form) orc_ntt_inv(ph, &c->ntt[j]);
    const uint64_t *c0 = ct + (size_t)j * n;
    for (size_t k = 0; k < n; k++) ph[k] = orc_add_mod(ph[k], c0[k], m->q);
  }
  free(spow); free(tmp);
}

int orc_bfv_decrypt(const orc_ctx *c, const uint64_t *ct, int size, uint64_t *plain) {
  /* [SEAL-recall: Decryptor::bfv_decrypt + RNSTool::decrypt_scale_and_round] */
  if (c->scheme != ORC_SCHEME_BFV || !c->sk_ntt) return -1;
  size_t n = c->n;
  int L = c->L;
  const orc_behz *b = c->behz;
  uint64_t *phase = (uint64_t *)malloc((size_t)L * n * 8);
  dot_ct_sk(c, ct, size, L, 0, phase);
  for (int j = 0; j < L; j++)
    for (size_t k = 0; k < n; k++)
      phase[(size_t)j * n + k] = orc_mulmod_b(phase[(size_t)j * n + k], b->tgamma_mod_q[j], &c->qmod[j]);
  uint64_t *tg = (uint64_t *)malloc((size_t)2 * n * 8);
  orc_bconv_apply(&b->q_to_tgamma, phase, tg, n);
  uint64_t t = c->t.q, gamma = b->gamma.q, gamma_half = gamma >> 1;
  for (size_t k = 0; k < n; k++) {
    uint64_t vt = orc_mul_mod(tg[k], b->neg_inv_q_mod_t, t);
    uint64_t vg = orc_mul_mod(tg[n + k], b->neg_inv_q_mod_gamma, gamma);
    uint64_t r;
    if (vg > gamma_half) r = orc_add_mod(vt, (gamma - vg) % t, t);
    else r = orc_sub_mod(vt, vg % t, t);
    if (r) r = orc_mul_mod(r, b->inv_gamma_mod_t, t);
    plain[k] = r;
  }
  free(tg); free(phase);
  return 0;
}

int orc_ckks_decrypt(const orc_ctx *c, const uint64_t *ct, int size, int nl, uint64_t *plain) {
  if (c->scheme != ORC_SCHEME_CKKS || !c->sk_ntt) return -1;
  dot_ct_sk(c, ct, size, nl, 1, plain);
  return 0;
}

/* [SEAL-recall: Decryptor::invariant_noise_budget] noise = |t * phase mod q| centred; budget =
 * bits(q) - bits(noise) - 1 */
int orc_bfv_noise_budget(const orc_ctx *c, const uint64_t *ct, int size) {
  if (c->scheme != ORC_SCHEME_BFV || !c->sk_ntt) return -1;
  size_t n = c->n;
  int L = c->L;
  uint64_t *phase = (uint64_t *)malloc((size_t)L * n * 8);
  dot_ct_sk(c, ct, size, L, 0, phase);
  /* compose t*phase mod q via Garner into long double magnitude (diagnostic only) */
  long double qprod = 1.0L;
  for (int j = 0; j < L; j++) qprod *= (long double)c->qmod[j].q;
  long double maxnorm = 0.0L;
  for (size_t k = 0; k < n; k++) {
    /* mixed-radix digits */
    uint64_t d[ORC_MAX_LIMBS];
    for (int j = 0; j < L; j++) {
      uint64_t qj = c->qmod[j].q;
      uint64_t v = orc_mul_mod(phase[(size_t)j * n + k], c->t.q % qj, qj);
      /* subtract lower digits */
      uint64_t acc = 0, rad = 1 % qj;
      for (int i = 0; i < j; i++) {
        acc = orc_add_mod(acc, orc_mul_mod(d[i] % qj, rad, qj), qj);
        rad = orc_mul_mod(rad, c->qmod[i].q % qj, qj);
      }
      d[j] = orc_mul_mod(orc_sub_mod(v, acc, qj), orc_inv_mod(rad, qj), qj);
    }
    long double val = 0.0L, rad = 1.0L;
    for (int j = 0; j < L; j++) { val += (long double)d[j] * rad; rad *= (long double)c->qmod[j].q; }
    if (val > qprod / 2) val = qprod - val;
    if (val > maxnorm) maxnorm = val;
  }
  free(phase);
  int qbits = (int)ceill(log2l(qprod));
  int nbits = maxnorm < 1 ? 0 : (int)floorl(log2l(maxnorm)) + 1;
  int budget = qbits - nbits - 1;
  return budget < 0 ? 0 : budget;
}
