/*
 * oracle/orc_eval.c -- TEST INFRASTRUCTURE (CPU oracle), see orc_internal.h header.
 *
 * seal::Evaluator restated.  Reference call sites (src/runtime/SealCiphertext.cpp):
 *   :92,:114   add / add_inplace                 -> orc_add
 *   :98,:118   sub / sub_inplace                 -> orc_sub
 *   :157,:193  negate / negate_inplace           -> orc_negate
 *   :104,:122  multiply / multiply_inplace (BFV) -> orc_bfv_multiply   (BEHZ)
 *   :105,:123,:160,:197 relinearize_inplace      -> orc_relinearize    (switch_key_inplace)
 *   :55,:60    rotate_rows(_inplace)             -> orc_rotate         (NAF + apply_galois + switch_key)
 *   :159,:196  multiply_plain(_inplace)          -> orc_bfv_multiply_plain
 *   :134,:175  add_plain(_inplace)               -> orc_bfv_add_plain
 *   :145,:184  sub_plain(_inplace)               -> orc_bfv_sub_plain
 * [SEAL-recall: evaluator.cpp (bfv_multiply, ckks_multiply, relinearize_internal, switch_key_inplace,
 *  rotate_internal, apply_galois_inplace, multiply_plain_normal, add_plain, sub_plain,
 *  rescale_to_next / mod_switch_drop_to_next), util/rns.cpp (fastbconv_m_tilde, sm_mrq, fast_floor,
 *  fastbconv_sk, divide_and_round_q_last_ntt_inplace), util/galois.cpp]
 */
#include "orc_internal.h"
#include "oracle.h"

#include <stdlib.h>
#include <string.h>
#include <time.h>


/* Scratch allocation.  Outside the timed loop of orc_time_mul_relin these are malloc / free.  Inside it they bump a
 * thread-local arena that was sized and touched before the clock started, so the CPU baseline does not time the C
 * library's allocator or first-touch page faults (SEAL itself draws its temporaries from a memory pool). */
static __thread unsigned char *g_arena = NULL;
static __thread size_t g_arena_size = 0, g_arena_used = 0;
static void *orc_salloc(size_t bytes) {
  if (!g_arena) return malloc(bytes);
  size_t need = (bytes + 63) & ~(size_t)63;
  if (g_arena_used + need > g_arena_size) return malloc(bytes); /* arena exhausted (sized from the scheme below): heap, freed by orc_sfree */
  void *p = g_arena + g_arena_used;
  g_arena_used += need;
  return p;
}
static void *orc_szalloc(size_t bytes) {
  void *p = orc_salloc(bytes);
  if (p) memset(p, 0, bytes);
  return p;
}
static void orc_sfree(void *p) {
  if (!g_arena || (unsigned char *)p < g_arena || (unsigned char *)p >= g_arena + g_arena_size) free(p); /* arena blocks are bump-allocated */
}

void orc_scaled_plain_addsub(const orc_ctx *c, const uint64_t *plain, uint64_t *poly, int sub);

void orc_ntt_forward(const orc_ctx *c, int i, uint64_t *a) { orc_ntt_fwd(a, &c->ntt[i]); }
void orc_ntt_inverse(const orc_ctx *c, int i, uint64_t *a) { orc_ntt_inv(a, &c->ntt[i]); }
void orc_ntt_forward_behz(const orc_ctx *c, int j, uint64_t *a) { orc_ntt_fwd(a, &c->behz->Bsk_ntt[j]); }
void orc_ntt_inverse_behz(const orc_ctx *c, int j, uint64_t *a) { orc_ntt_inv(a, &c->behz->Bsk_ntt[j]); }
void orc_ntt_forward_plain(const orc_ctx *c, uint64_t *a) { orc_ntt_fwd(a, c->t_ntt); }
void orc_ntt_inverse_plain(const orc_ctx *c, uint64_t *a) { orc_ntt_inv(a, c->t_ntt); }

/* ---------- element-wise ---------- */
int orc_add(const orc_ctx *c, const uint64_t *a, const uint64_t *b, int size, int nl, uint64_t *out) {
  size_t n = c->n;
  for (int p = 0; p < size; p++)
    for (int j = 0; j < nl; j++) {
      uint64_t q = c->qmod[j].q;
      size_t o = ((size_t)p * nl + j) * n;
      for (size_t k = 0; k < n; k++) out[o + k] = orc_add_mod(a[o + k], b[o + k], q);
    }
  return 0;
}
int orc_sub(const orc_ctx *c, const uint64_t *a, const uint64_t *b, int size, int nl, uint64_t *out) {
  size_t n = c->n;
  for (int p = 0; p < size; p++)
    for (int j = 0; j < nl; j++) {
      uint64_t q = c->qmod[j].q;
      size_t o = ((size_t)p * nl + j) * n;
      for (size_t k = 0; k < n; k++) out[o + k] = orc_sub_mod(a[o + k], b[o + k], q);
    }
  return 0;
}
int orc_negate(const orc_ctx *c, const uint64_t *a, int size, int nl, uint64_t *out) {
  size_t n = c->n;
  for (int p = 0; p < size; p++)
    for (int j = 0; j < nl; j++) {
      uint64_t q = c->qmod[j].q;
      size_t o = ((size_t)p * nl + j) * n;
      for (size_t k = 0; k < n; k++) out[o + k] = orc_neg_mod(a[o + k], q);
    }
  return 0;
}

/* ---------- key switching ---------- */
/* [SEAL-recall: Evaluator::switch_key_inplace]
 * target: [nl][n] polynomial to switch (BFV: coefficient form, CKKS: NTT form), at a level with nl
 * data limbs; key: [L][2][nkey][n]; out2: [2][nl][n] receives ONLY the key-switch contribution
 * (the caller adds it to c0/c1), in the ciphertext's own form. */
int orc_keyswitch(const orc_ctx *c, const uint64_t *target, int nl, const uint64_t *key, uint64_t *out2) {
  size_t n = c->n;
  int K = c->nkey;
  int ckks = (c->scheme == ORC_SCHEME_CKKS);
  int rns = nl + 1;
  uint64_t *tcoef = (uint64_t *)orc_salloc((size_t)nl * n * 8);
  memcpy(tcoef, target, (size_t)nl * n * 8);
  if (ckks)
    for (int j = 0; j < nl; j++) orc_ntt_inv(tcoef + (size_t)j * n, &c->ntt[j]);
  uint64_t *prod = (uint64_t *)orc_szalloc((size_t)2 * rns * n * 8); /* [2][rns][n] */
  uint64_t *tntt = (uint64_t *)orc_salloc(n * 8);
  u128 *acc = (u128 *)orc_salloc((size_t)2 * n * sizeof(u128));
  for (int I = 0; I < rns; I++) {
    int ki = (I == nl) ? K - 1 : I;
    const orc_mod *mk = &c->qmod[ki];
    memset(acc, 0, (size_t)2 * n * sizeof(u128));
    for (int J = 0; J < nl; J++) {
      const uint64_t *op;
      if (ckks && I == J) {
        op = target + (size_t)J * n;
      } else {
        const uint64_t *src = tcoef + (size_t)J * n;
        if (c->qmod[J].q <= mk->q) memcpy(tntt, src, n * 8);
        else for (size_t k = 0; k < n; k++) tntt[k] = orc_barrett64(src[k], mk);
        orc_ntt_fwd(tntt, &c->ntt[ki]);
        op = tntt;
      }
      for (int comp = 0; comp < 2; comp++) {
        const uint64_t *kk = key + (((size_t)J * 2 + comp) * K + ki) * n;
        u128 *a = acc + (size_t)comp * n;
        for (size_t k = 0; k < n; k++) a[k] += (u128)op[k] * kk[k]; /* lazy: <= 16 summands of < 2^124 */
      }
    }
    for (int comp = 0; comp < 2; comp++) {
      uint64_t *dst = prod + ((size_t)comp * rns + I) * n;
      const u128 *a = acc + (size_t)comp * n;
      for (size_t k = 0; k < n; k++) dst[k] = orc_barrett128(a[k], mk);
    }
  }
  /* mod-down by the special prime with rounding */
  const orc_mod *msp = &c->qmod[K - 1];
  uint64_t qk = msp->q, half = qk >> 1;
  for (int comp = 0; comp < 2; comp++) {
    uint64_t *last = prod + ((size_t)comp * rns + nl) * n;
    orc_ntt_inv(last, &c->ntt[K - 1]);
    for (size_t k = 0; k < n; k++) last[k] = orc_barrett64(last[k] + half, msp);
    for (int j = 0; j < nl; j++) {
      const orc_mod *mj = &c->qmod[j];
      uint64_t qj = mj->q;
      uint64_t fix = qj - orc_barrett64(half, mj);
      for (size_t k = 0; k < n; k++) tntt[k] = orc_add_mod(orc_barrett64(last[k], mj), fix % qj, qj);
      uint64_t *pj = prod + ((size_t)comp * rns + j) * n;
      if (ckks) orc_ntt_fwd(tntt, &c->ntt[j]);
      else orc_ntt_inv(pj, &c->ntt[j]);
      uint64_t *dst = out2 + ((size_t)comp * nl + j) * n;
      for (size_t k = 0; k < n; k++)
        dst[k] = orc_mulmod_b(orc_sub_mod(pj[k], tntt[k], qj), c->inv_special_mod_q[j], mj);
    }
  }
  orc_sfree(acc); orc_sfree(tntt); orc_sfree(prod); orc_sfree(tcoef);
  return 0;
}

int orc_relinearize(const orc_ctx *c, const uint64_t *ct3, int nl, uint64_t *out2) {
  /* [SEAL-recall: Evaluator::relinearize_internal, size 3 -> 2] */
  size_t n = c->n;
  if (!c->relin) return -1;
  uint64_t *ks = (uint64_t *)orc_salloc((size_t)2 * nl * n * 8);
  orc_keyswitch(c, ct3 + (size_t)2 * nl * n, nl, c->relin, ks);
  orc_add(c, ct3, ks, 2, nl, out2);
  orc_sfree(ks);
  return 0;
}

/* ---------- Galois ---------- */
int orc_galois_permute(const orc_ctx *c, const uint64_t *poly, int nl, uint32_t elt, int ntt_form, uint64_t *out) {
  size_t n = c->n;
  int logn = c->logn;
  for (int j = 0; j < nl; j++) {
    const uint64_t *in = poly + (size_t)j * n;
    uint64_t *o = out + (size_t)j * n;
    uint64_t q = c->qmod[j].q;
    if (ntt_form) {
      /* [SEAL-recall: GaloisTool::apply_galois_ntt] */
      for (size_t i = 0; i < n; i++) {
        uint32_t rev = orc_bitrev((uint32_t)(i + n), logn + 1);
        uint64_t idx = (((uint64_t)elt * rev) >> 1) & (n - 1);
        o[i] = in[orc_bitrev((uint32_t)idx, logn)];
      }
    } else {
      /* [SEAL-recall: GaloisTool::apply_galois] x^i -> x^(i*elt) with sign flip on wrap */
      uint64_t raw = 0;
      for (size_t i = 0; i < n; i++, raw += elt) {
        uint64_t idx = raw & (n - 1);
        uint64_t v = in[i];
        if ((raw >> logn) & 1) v = orc_neg_mod(v, q);
        o[idx] = v;
      }
    }
  }
  return 0;
}

int orc_apply_galois(const orc_ctx *c, const uint64_t *ct, int nl, uint32_t elt, uint64_t *out2) {
  /* [SEAL-recall: Evaluator::apply_galois_inplace] (c0,c1) -> (g(c0) + ks0, ks1), ks = KS(g(c1)) */
  size_t n = c->n;
  int ntt_form = (c->scheme == ORC_SCHEME_CKKS);
  const uint64_t *key = NULL;
  for (int i = 0; i < c->ngal; i++) if (c->gal_elt[i] == elt) key = c->gal_key[i];
  if (!key) return -1;
  size_t pw = (size_t)nl * n;
  uint64_t *g0 = (uint64_t *)orc_salloc(pw * 8), *g1 = (uint64_t *)orc_salloc(pw * 8);
  uint64_t *ks = (uint64_t *)orc_salloc(2 * pw * 8);
  orc_galois_permute(c, ct, nl, elt, ntt_form, g0);
  orc_galois_permute(c, ct + pw, nl, elt, ntt_form, g1);
  orc_keyswitch(c, g1, nl, key, ks);
  orc_add(c, g0, ks, 1, nl, out2);
  memcpy(out2 + pw, ks + pw, pw * 8);
  orc_sfree(ks); orc_sfree(g1); orc_sfree(g0);
  return 0;
}

int orc_rotate(const orc_ctx *c, const uint64_t *ct, int nl, int steps, uint64_t *out2) {
  /* [SEAL-recall: Evaluator::rotate_internal] direct key if present, else NAF decomposition */
  size_t words = (size_t)2 * nl * c->n;
  if (steps == 0) { memcpy(out2, ct, words * 8); return 0; }
  uint32_t elt = orc_galois_elt_from_step(c, steps);
  if (!elt) return -2;
  for (int i = 0; i < c->ngal; i++)
    if (c->gal_elt[i] == elt) return orc_apply_galois(c, ct, nl, elt, out2);
  int naf[40];
  int cnt = orc_naf(steps, naf);
  if (cnt == 1) return -3; /* "Galois key not present" */
  uint64_t *cur = (uint64_t *)orc_salloc(words * 8), *nxt = (uint64_t *)orc_salloc(words * 8);
  memcpy(cur, ct, words * 8);
  int rc = 0;
  for (int i = 0; i < cnt && !rc; i++) {
    if ((size_t)abs(naf[i]) == (c->n >> 1)) continue;
    rc = orc_rotate(c, cur, nl, naf[i], nxt);
    uint64_t *t = cur; cur = nxt; nxt = t;
  }
  memcpy(out2, cur, words * 8);
  orc_sfree(cur); orc_sfree(nxt);
  return rc;
}

/* ---------- BFV multiply (BEHZ) ---------- */
static void behz_extend(const orc_ctx *c, const uint64_t *poly /*[L][n] coeff*/, uint64_t *q_ntt /*[L][n]*/,
                        uint64_t *bsk_ntt /*[nBsk][n]*/) {
  /* steps (1)-(3): NTT copy in base q; q -> Bsk u {m_tilde}; Montgomery reduce; NTT in Bsk */
  const orc_behz *b = c->behz;
  size_t n = c->n;
  int L = c->L, nBsk = b->nBsk;
  memcpy(q_ntt, poly, (size_t)L * n * 8);
  for (int j = 0; j < L; j++) orc_ntt_fwd(q_ntt + (size_t)j * n, &c->ntt[j]);
  /* fastbconv_m_tilde */
  uint64_t *tmp = (uint64_t *)orc_salloc((size_t)L * n * 8);
  for (int j = 0; j < L; j++)
    for (size_t k = 0; k < n; k++)
      tmp[(size_t)j * n + k] = orc_mulmod_b(poly[(size_t)j * n + k], b->mtilde_mod_q[j], &c->qmod[j]);
  uint64_t *ext = (uint64_t *)orc_salloc((size_t)(nBsk + 1) * n * 8);
  orc_bconv_apply(&b->q_to_Bsk, tmp, ext, n);
  orc_bconv_apply(&b->q_to_mtilde, tmp, ext + (size_t)nBsk * n, n);
  /* sm_mrq */
  uint64_t mt = b->m_tilde.q, mt_half = mt >> 1;
  const uint64_t *in_mt = ext + (size_t)nBsk * n;
  for (int j = 0; j < nBsk; j++) {
    const orc_mod *pj = &b->Bsk[j];
    for (size_t k = 0; k < n; k++) {
      uint64_t r = (in_mt[k] * b->neg_inv_q_mod_mtilde) & (mt - 1);
      if (r >= mt_half) r += pj->q - mt;
      uint64_t v = orc_add_mod(orc_mulmod_b(r, b->q_mod_Bsk[j], pj), ext[(size_t)j * n + k], pj->q);
      bsk_ntt[(size_t)j * n + k] = orc_mulmod_b(v, b->inv_mtilde_mod_Bsk[j], pj);
    }
    orc_ntt_fwd(bsk_ntt + (size_t)j * n, &b->Bsk_ntt[j]);
  }
  orc_sfree(ext); orc_sfree(tmp);
}

int orc_bfv_multiply(const orc_ctx *c, const uint64_t *a, const uint64_t *bb, uint64_t *out3) {
  /* [SEAL-recall: Evaluator::bfv_multiply], size-2 x size-2 -> size-3, coefficient form */
  if (c->scheme != ORC_SCHEME_BFV) return -1;
  const orc_behz *b = c->behz;
  size_t n = c->n;
  int L = c->L, nBsk = b->nBsk;
  size_t qw = (size_t)L * n, bw = (size_t)nBsk * n;
  uint64_t *aq = (uint64_t *)orc_salloc(2 * qw * 8), *bq = (uint64_t *)orc_salloc(2 * qw * 8);
  uint64_t *aB = (uint64_t *)orc_salloc(2 * bw * 8), *bB = (uint64_t *)orc_salloc(2 * bw * 8);
  for (int p = 0; p < 2; p++) {
    behz_extend(c, a + p * qw, aq + p * qw, aB + p * bw);
    behz_extend(c, bb + p * qw, bq + p * qw, bB + p * bw);
  }
  uint64_t *dq = (uint64_t *)orc_salloc(3 * qw * 8), *dB = (uint64_t *)orc_salloc(3 * bw * 8);
  /* step (4): dyadic tensor in both bases; step (5): inverse NTT */
  for (int j = 0; j < L; j++) {
    const orc_mod *m = &c->qmod[j];
    size_t o = (size_t)j * n;
    for (size_t k = 0; k < n; k++) {
      uint64_t a0 = aq[o + k], a1 = aq[qw + o + k], b0 = bq[o + k], b1 = bq[qw + o + k];
      dq[o + k] = orc_mulmod_b(a0, b0, m);
      dq[qw + o + k] = orc_add_mod(orc_mulmod_b(a0, b1, m), orc_mulmod_b(a1, b0, m), m->q);
      dq[2 * qw + o + k] = orc_mulmod_b(a1, b1, m);
    }
    for (int p = 0; p < 3; p++) orc_ntt_inv(dq + p * qw + o, &c->ntt[j]);
  }
  for (int j = 0; j < nBsk; j++) {
    const orc_mod *m = &b->Bsk[j];
    size_t o = (size_t)j * n;
    for (size_t k = 0; k < n; k++) {
      uint64_t a0 = aB[o + k], a1 = aB[bw + o + k], b0 = bB[o + k], b1 = bB[bw + o + k];
      dB[o + k] = orc_mulmod_b(a0, b0, m);
      dB[bw + o + k] = orc_add_mod(orc_mulmod_b(a0, b1, m), orc_mulmod_b(a1, b0, m), m->q);
      dB[2 * bw + o + k] = orc_mulmod_b(a1, b1, m);
    }
    for (int p = 0; p < 3; p++) orc_ntt_inv(dB + p * bw + o, &b->Bsk_ntt[j]);
  }
  /* steps (6)-(8) per output polynomial */
  uint64_t t = c->t.q;
  uint64_t *conv = (uint64_t *)orc_salloc(bw * 8), *fl = (uint64_t *)orc_salloc(bw * 8);
  uint64_t *msk = (uint64_t *)orc_salloc(n * 8);
  for (int p = 0; p < 3; p++) {
    uint64_t *pq = dq + p * qw, *pB = dB + p * bw;
    for (int j = 0; j < L; j++)
      for (size_t k = 0; k < n; k++) pq[(size_t)j * n + k] = orc_mulmod_b(pq[(size_t)j * n + k], t % c->qmod[j].q, &c->qmod[j]);
    for (int j = 0; j < nBsk; j++)
      for (size_t k = 0; k < n; k++) pB[(size_t)j * n + k] = orc_mulmod_b(pB[(size_t)j * n + k], t % b->Bsk[j].q, &b->Bsk[j]);
    /* fast_floor: (x_Bsk - conv_{q->Bsk}(x_q)) * q^-1 */
    orc_bconv_apply(&b->q_to_Bsk, pq, conv, n);
    for (int j = 0; j < nBsk; j++) {
      const orc_mod *m = &b->Bsk[j];
      for (size_t k = 0; k < n; k++)
        fl[(size_t)j * n + k] = orc_mulmod_b(orc_sub_mod(pB[(size_t)j * n + k], conv[(size_t)j * n + k], m->q), b->inv_q_mod_Bsk[j], m);
    }
    /* fastbconv_sk: B -> q with Shenoy-Kumaresan correction through m_sk */
    uint64_t *dst = out3 + p * qw;
    orc_bconv_apply(&b->B_to_q, fl, dst, n);
    orc_bconv_apply(&b->B_to_msk, fl, msk, n);
    uint64_t qs = b->m_sk.q, qs_half = qs >> 1;
    const uint64_t *in_sk = fl + (size_t)b->nB * n;
    for (size_t k = 0; k < n; k++) {
      uint64_t alpha = orc_mulmod_b(orc_sub_mod(msk[k], in_sk[k], qs), b->inv_B_mod_msk, &b->m_sk);
      for (int j = 0; j < L; j++) {
        const orc_mod *m = &c->qmod[j];
        uint64_t *d = dst + (size_t)j * n + k;
        if (alpha > qs_half)
          *d = orc_add_mod(*d, orc_mulmod_b(orc_barrett64(qs - alpha, m), b->B_mod_q[j], m), m->q);
        else
          *d = orc_sub_mod(*d, orc_mulmod_b(orc_barrett64(alpha, m), b->B_mod_q[j], m), m->q);
      }
    }
  }
  orc_sfree(msk); orc_sfree(fl); orc_sfree(conv); orc_sfree(dB); orc_sfree(dq); orc_sfree(bB); orc_sfree(aB); orc_sfree(bq); orc_sfree(aq);
  return 0;
}

int orc_bfv_mul_relin(const orc_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out2) {
  /* SealCiphertext::multiply = multiply + relinearize_inplace (src/runtime/SealCiphertext.cpp:102-107) */
  uint64_t *t3 = (uint64_t *)orc_salloc((size_t)3 * c->L * c->n * 8);
  int rc = orc_bfv_multiply(c, a, b, t3);
  if (!rc) rc = orc_relinearize(c, t3, c->L, out2);
  orc_sfree(t3);
  return rc;
}

/* ---------- BFV plain ops ---------- */
int orc_bfv_multiply_plain(const orc_ctx *c, const uint64_t *ct, int size, const uint64_t *plain, uint64_t *out) {
  /* [SEAL-recall: Evaluator::multiply_plain_normal] lift plain to each q_i (upper half gets +q_i-t),
   * NTT, dyadic multiply, INTT.  Residues are canonical, so SEAL's mono-coefficient shortcut agrees. */
  size_t n = c->n;
  int L = c->L;
  uint64_t *pl = (uint64_t *)orc_salloc(n * 8), *tmp = (uint64_t *)orc_salloc(n * 8);
  for (int j = 0; j < L; j++) {
    const orc_mod *m = &c->qmod[j];
    for (size_t k = 0; k < n; k++)
      pl[k] = plain[k] + (plain[k] >= c->upper_half_threshold ? c->upper_half_increment[j] : 0);
    orc_ntt_fwd(pl, &c->ntt[j]);
    for (int p = 0; p < size; p++) {
      size_t o = ((size_t)p * L + j) * n;
      memcpy(tmp, ct + o, n * 8);
      orc_ntt_fwd(tmp, &c->ntt[j]);
      for (size_t k = 0; k < n; k++) tmp[k] = orc_mulmod_b(tmp[k], pl[k], m);
      orc_ntt_inv(tmp, &c->ntt[j]);
      memcpy(out + o, tmp, n * 8);
    }
  }
  orc_sfree(tmp); orc_sfree(pl);
  return 0;
}
int orc_bfv_add_plain(const orc_ctx *c, const uint64_t *ct, int size, const uint64_t *plain, uint64_t *out) {
  memmove(out, ct, (size_t)size * c->L * c->n * 8);
  orc_scaled_plain_addsub(c, plain, out, 0);
  return 0;
}
int orc_bfv_sub_plain(const orc_ctx *c, const uint64_t *ct, int size, const uint64_t *plain, uint64_t *out) {
  memmove(out, ct, (size_t)size * c->L * c->n * 8);
  orc_scaled_plain_addsub(c, plain, out, 1);
  return 0;
}

/* ---------- CKKS ---------- */
int orc_ckks_multiply(const orc_ctx *c, const uint64_t *a, const uint64_t *b, int nl, uint64_t *out3) {
  /* [SEAL-recall: Evaluator::ckks_multiply] dyadic tensor in NTT form */
  size_t n = c->n, pw = (size_t)nl * n;
  for (int j = 0; j < nl; j++) {
    const orc_mod *m = &c->qmod[j];
    size_t o = (size_t)j * n;
    for (size_t k = 0; k < n; k++) {
      uint64_t a0 = a[o + k], a1 = a[pw + o + k], b0 = b[o + k], b1 = b[pw + o + k];
      out3[o + k] = orc_mulmod_b(a0, b0, m);
      out3[pw + o + k] = orc_add_mod(orc_mulmod_b(a0, b1, m), orc_mulmod_b(a1, b0, m), m->q);
      out3[2 * pw + o + k] = orc_mulmod_b(a1, b1, m);
    }
  }
  return 0;
}
int orc_ckks_mul_relin(const orc_ctx *c, const uint64_t *a, const uint64_t *b, int nl, uint64_t *out2) {
  uint64_t *t3 = (uint64_t *)orc_salloc((size_t)3 * nl * c->n * 8);
  orc_ckks_multiply(c, a, b, nl, t3);
  int rc = orc_relinearize(c, t3, nl, out2);
  orc_sfree(t3);
  return rc;
}
int orc_ckks_rescale(const orc_ctx *c, const uint64_t *ct, int size, int nl, uint64_t *out) {
  /* [SEAL-recall: RNSTool::divide_and_round_q_last_ntt_inplace] drop limb nl-1 with rounding */
  if (nl < 2) return -1;
  size_t n = c->n;
  int last = nl - 1;
  const orc_mod *ml = &c->qmod[last];
  uint64_t half = ml->q >> 1;
  uint64_t *li = (uint64_t *)orc_salloc(n * 8), *tmp = (uint64_t *)orc_salloc(n * 8);
  for (int p = 0; p < size; p++) {
    memcpy(li, ct + ((size_t)p * nl + last) * n, n * 8);
    orc_ntt_inv(li, &c->ntt[last]);
    for (size_t k = 0; k < n; k++) li[k] = orc_add_mod(li[k], half, ml->q);
    for (int j = 0; j < last; j++) {
      const orc_mod *mj = &c->qmod[j];
      uint64_t neg_half = mj->q - orc_barrett64(half, mj);
      for (size_t k = 0; k < n; k++) tmp[k] = orc_add_mod(orc_barrett64(li[k], mj), neg_half % mj->q, mj->q);
      orc_ntt_fwd(tmp, &c->ntt[j]);
      const uint64_t *src = ct + ((size_t)p * nl + j) * n;
      uint64_t *dst = out + ((size_t)p * last + j) * n;
      for (size_t k = 0; k < n; k++)
        dst[k] = orc_mulmod_b(orc_sub_mod(src[k], tmp[k], mj->q), c->inv_qlast_mod_q[last][j], mj);
    }
  }
  orc_sfree(tmp); orc_sfree(li);
  return 0;
}
int orc_ckks_mod_switch(const orc_ctx *c, const uint64_t *ct, int size, int nl, uint64_t *out) {
  /* [SEAL-recall: Evaluator::mod_switch_drop_to_next] CKKS: simply drop the last limb */
  if (nl < 2) return -1;
  size_t n = c->n;
  for (int p = 0; p < size; p++)
    for (int j = 0; j < nl - 1; j++)
      memmove(out + ((size_t)p * (nl - 1) + j) * n, ct + ((size_t)p * nl + j) * n, n * 8);
  return 0;
}
int orc_ckks_multiply_plain(const orc_ctx *c, const uint64_t *ct, int size, int nl, const uint64_t *plain, uint64_t *out) {
  size_t n = c->n;
  for (int p = 0; p < size; p++)
    for (int j = 0; j < nl; j++) {
      size_t o = ((size_t)p * nl + j) * n;
      for (size_t k = 0; k < n; k++) out[o + k] = orc_mulmod_b(ct[o + k], plain[(size_t)j * n + k], &c->qmod[j]);
    }
  return 0;
}
int orc_ckks_add_plain(const orc_ctx *c, const uint64_t *ct, int size, int nl, const uint64_t *plain, uint64_t *out) {
  size_t n = c->n;
  memmove(out, ct, (size_t)size * nl * n * 8);
  for (int j = 0; j < nl; j++)
    for (size_t k = 0; k < n; k++) {
      size_t o = (size_t)j * n + k;
      out[o] = orc_add_mod(out[o], plain[o], c->qmod[j].q);
    }
  return 0;
}

/* ---------- CPU baseline timing ---------- */
double orc_time_mul_relin(const orc_ctx *c, const uint64_t *a, const uint64_t *b, int nl, int iters, uint64_t *out2) {
  struct timespec t0, t1;
  /* arena: every temporary of one mul+relin, touched once before timing.  Nothing is freed inside the arena, so it is sized by the
   * sum of all temporaries: CKKS ~ 3L + L(L+1)-independent key-switch buffers (about 40 limbs at L = 4); BFV adds the BEHZ
   * operands and products in q and Bsk (about 300 limbs at L = 8, nBsk = 9 or 10) */
  size_t L = (size_t)(nl > 0 ? nl : c->L);
  size_t nbsk = c->behz ? (size_t)c->behz->nBsk : 0;
  size_t limbs = 64 + 16 * L + 12 * nbsk + 2 * (L + nbsk + 1) * 4;
  if (limbs < 256) limbs = 256;
  size_t bytes = limbs * c->n * 8;
  unsigned char *arena = (unsigned char *)malloc(bytes);
  if (arena) {
    memset(arena, 0, bytes);
    g_arena = arena; g_arena_size = bytes;
  }
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int i = 0; i < iters; i++) {
    g_arena_used = 0;
    if (c->scheme == ORC_SCHEME_CKKS) orc_ckks_mul_relin(c, a, b, nl, out2);
    else orc_bfv_mul_relin(c, a, b, out2);
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  g_arena = NULL; g_arena_size = g_arena_used = 0;
  free(arena);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
