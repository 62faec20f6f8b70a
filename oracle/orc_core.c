/*
 * oracle/orc_core.c -- TEST INFRASTRUCTURE (CPU oracle), see orc_internal.h header.
 *
 * Number theory + negacyclic NTT.  Restates [SEAL-recall] util/numth.cpp (is_prime,
 * get_primes, try_minimal_primitive_root), util/ntt.cpp + util/dwthandler.h (Harvey lazy
 * butterflies, root powers in bit-reversed order).  Used by the reference through
 * seal::Evaluator / BatchEncoder at src/runtime/SealCiphertext.cpp:104-105,122-123,159,196
 * and src/runtime/SealCiphertextFactory.cpp:130,151.
 */
#include "orc_internal.h"

#include <stdlib.h>
#include <string.h>

void orc_mod_init(orc_mod *m, uint64_t q) {
  m->q = q;
  /* floor(2^128 / q) computed by long division of (2^128 - 1) / q; identical unless q | 2^128,
   * which happens only for powers of two where we correct explicitly. */
  u128 all1 = ~(u128)0;
  u128 quo = all1 / q;
  if ((q & (q - 1)) == 0) quo += 1; /* q = 2^k divides 2^128 exactly (m_tilde = 2^32) */
  m->ratio[0] = (uint64_t)quo;
  m->ratio[1] = (uint64_t)(quo >> 64);
}

uint64_t orc_pow_mod(uint64_t b, uint64_t e, uint64_t q) {
  uint64_t r = 1 % q;
  b %= q;
  while (e) {
    if (e & 1) r = orc_mul_mod(r, b, q);
    b = orc_mul_mod(b, b, q);
    e >>= 1;
  }
  return r;
}

uint64_t orc_inv_mod(uint64_t a, uint64_t q) {
  /* extended Euclid on signed 128-bit to be safe for 64-bit moduli */
  __int128 t = 0, newt = 1;
  __int128 r = q, newr = a % q;
  while (newr != 0) {
    __int128 quo = r / newr;
    __int128 tmp = t - quo * newt; t = newt; newt = tmp;
    tmp = r - quo * newr; r = newr; newr = tmp;
  }
  if (r != 1) return 0; /* not invertible */
  if (t < 0) t += q;
  return (uint64_t)t;
}

/* Deterministic Miller-Rabin for 64-bit integers. */
int orc_is_prime(uint64_t n) {
  if (n < 2) return 0;
  static const uint64_t small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
  for (size_t i = 0; i < sizeof(small) / sizeof(small[0]); i++) {
    if (n == small[i]) return 1;
    if (n % small[i] == 0) return 0;
  }
  uint64_t d = n - 1;
  int r = 0;
  while ((d & 1) == 0) { d >>= 1; r++; }
  for (size_t i = 0; i < sizeof(small) / sizeof(small[0]); i++) {
    uint64_t a = small[i];
    uint64_t x = orc_pow_mod(a, d, n);
    if (x == 1 || x == n - 1) continue;
    int comp = 1;
    for (int j = 1; j < r; j++) {
      x = orc_mul_mod(x, x, n);
      if (x == n - 1) { comp = 0; break; }
    }
    if (comp) return 0;
  }
  return 1;
}

/* [SEAL-recall: util::get_primes] primes p = 1 (mod 2*ntt_size) with exactly bit_size bits,
 * in DEcreasing order starting from 2^bit_size - 2*ntt_size + 1. */
int orc_get_primes(size_t ntt_size, int bit_size, int count, uint64_t *out) {
  uint64_t factor = 2 * (uint64_t)ntt_size;
  uint64_t value = ((uint64_t)1 << bit_size) - factor + 1;
  uint64_t lower = (uint64_t)1 << (bit_size - 1);
  int got = 0;
  while (got < count && value > lower) {
    if (orc_is_prime(value)) out[got++] = value;
    value -= factor;
  }
  return got == count ? 0 : -1;
}

/* [SEAL-recall: util::try_minimal_primitive_root] smallest primitive two_n-th root mod q. */
int orc_minimal_primitive_root(uint64_t two_n, uint64_t q, uint64_t *root) {
  if ((q - 1) % two_n) return -1;
  uint64_t e = (q - 1) / two_n;
  uint64_t g = 0;
  for (uint64_t c = 2; c < q; c++) {
    g = orc_pow_mod(c, e, q);
    /* primitive iff g^(two_n/2) == -1 */
    if (orc_pow_mod(g, two_n / 2, q) == q - 1) break;
    g = 0;
  }
  if (!g) return -1;
  uint64_t gsq = orc_mul_mod(g, g, q);
  uint64_t cur = g, best = g;
  for (uint64_t i = 0; i < two_n / 2; i++) { /* all odd powers of g */
    if (cur < best) best = cur;
    cur = orc_mul_mod(cur, gsq, q);
  }
  *root = best;
  return 0;
}

uint32_t orc_bitrev(uint32_t x, int bits) {
  uint32_t r = 0;
  for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
  return r;
}

int orc_ntt_init(orc_ntt *t, int logn, uint64_t q) {
  memset(t, 0, sizeof(*t));
  t->logn = logn;
  t->n = (size_t)1 << logn;
  orc_mod_init(&t->mod, q);
  if (orc_minimal_primitive_root(2 * (uint64_t)t->n, q, &t->root)) return -1;
  size_t n = t->n;
  t->tw = (uint64_t *)malloc(4 * n * sizeof(uint64_t));
  t->twq = t->tw + n; t->itw = t->tw + 2 * n; t->itwq = t->tw + 3 * n;
  /* powers of root in natural order, then scatter to bit-reversed index */
  uint64_t p = 1;
  for (size_t i = 0; i < n; i++) {
    size_t j = orc_bitrev((uint32_t)i, logn);
    t->tw[j] = p;
    p = orc_mul_mod(p, t->root, q);
  }
  for (size_t i = 0; i < n; i++) {
    t->twq[i] = orc_shoup(t->tw[i], q);
    t->itw[i] = orc_inv_mod(t->tw[i], q);
    t->itwq[i] = orc_shoup(t->itw[i], q);
  }
  t->inv_n = orc_inv_mod((uint64_t)n % q, q);
  t->inv_nq = orc_shoup(t->inv_n, q);
  return 0;
}

void orc_ntt_free(orc_ntt *t) { free(t->tw); t->tw = NULL; }

/* Forward negacyclic NTT, Cooley-Tukey, natural order in -> bit-reversed order out.
 * Harvey lazy butterflies: values stay in [0,4q), final pass reduces to [0,q).
 * [SEAL-recall: DWTHandler::transform_to_rev + ntt_negacyclic_harvey] */
void orc_ntt_fwd(uint64_t *a, const orc_ntt *t) {
  const uint64_t q = t->mod.q, two_q = 2 * q;
  size_t n = t->n;
  size_t gap = n >> 1;
  for (size_t m = 1; m < n; m <<= 1, gap >>= 1) {
    for (size_t i = 0; i < m; i++) {
      uint64_t w = t->tw[m + i], wq = t->twq[m + i];
      uint64_t *x = a + 2 * i * gap, *y = x + gap;
      for (size_t j = 0; j < gap; j++) {
        uint64_t u = x[j] >= two_q ? x[j] - two_q : x[j];
        uint64_t v = orc_mul_shoup_lazy(y[j], w, wq, q);
        x[j] = u + v;
        y[j] = u + two_q - v;
      }
    }
  }
  for (size_t i = 0; i < n; i++) {
    uint64_t v = a[i];
    if (v >= two_q) v -= two_q;
    if (v >= q) v -= q;
    a[i] = v;
  }
}

/* Inverse negacyclic NTT, Gentleman-Sande, bit-reversed in -> natural out, scaled by n^-1.
 * Accepts inputs in [0,2q); output in [0,q).
 * [SEAL-recall: DWTHandler::transform_from_rev + inverse_ntt_negacyclic_harvey] */
void orc_ntt_inv(uint64_t *a, const orc_ntt *t) {
  const uint64_t q = t->mod.q, two_q = 2 * q;
  size_t n = t->n;
  size_t gap = 1;
  for (size_t m = n >> 1; m >= 1; m >>= 1, gap <<= 1) {
    for (size_t i = 0; i < m; i++) {
      uint64_t w = t->itw[m + i], wq = t->itwq[m + i];
      uint64_t *x = a + 2 * i * gap, *y = x + gap;
      for (size_t j = 0; j < gap; j++) {
        uint64_t u = x[j], v = y[j];
        uint64_t s = u + v;
        x[j] = s >= two_q ? s - two_q : s;
        y[j] = orc_mul_shoup_lazy(u + two_q - v, w, wq, q);
      }
    }
  }
  for (size_t i = 0; i < n; i++) {
    uint64_t v = orc_mul_shoup_lazy(a[i], t->inv_n, t->inv_nq, q);
    a[i] = v >= q ? v - q : v;
  }
}

/* ---- fast base conversion [SEAL-recall: BaseConverter::fast_convert_array] ----
 * out_j = sum_i ( in_i * (Q/q_i)^-1 mod q_i ) * ((Q/q_i) mod p_j)  mod p_j   (no correction) */
void orc_bconv_init(orc_bconv *c, const orc_mod *in, int nin, const orc_mod *out, int nout) {
  memset(c, 0, sizeof(*c));
  c->nin = nin; c->nout = nout;
  for (int i = 0; i < nin; i++) c->in[i] = in[i];
  for (int j = 0; j < nout; j++) c->out[j] = out[j];
  for (int i = 0; i < nin; i++) {
    uint64_t qi = in[i].q, p = 1 % qi;
    for (int k = 0; k < nin; k++) if (k != i) p = orc_mul_mod(p, in[k].q % qi, qi);
    c->inv_punct[i] = orc_inv_mod(p, qi);
    for (int j = 0; j < nout; j++) {
      uint64_t pj = out[j].q, v = 1 % pj;
      for (int k = 0; k < nin; k++) if (k != i) v = orc_mul_mod(v, in[k].q % pj, pj);
      c->mat[j][i] = v;
    }
  }
}

void orc_bconv_apply(const orc_bconv *c, const uint64_t *in, uint64_t *out, size_t n) {
  for (size_t k = 0; k < n; k++) {
    uint64_t tmp[ORC_MAX_LIMBS];
    for (int i = 0; i < c->nin; i++) tmp[i] = orc_mulmod_b(in[(size_t)i * n + k], c->inv_punct[i], &c->in[i]);
    for (int j = 0; j < c->nout; j++) {
      uint64_t pj = c->out[j].q;
      uint64_t acc = 0;
      for (int i = 0; i < c->nin; i++) acc = orc_add_mod(acc, orc_mul_mod(tmp[i] % pj, c->mat[j][i], pj), pj);
      out[(size_t)j * n + k] = acc;
    }
  }
}

/* ---- sampler: splitmix64-seeded xoshiro256** (this repo's sampling spec) ---- */
static uint64_t splitmix64(uint64_t *x) {
  uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
void orc_rng_seed(orc_rng *r, uint64_t seed) {
  uint64_t x = seed;
  for (int i = 0; i < 4; i++) r->s[i] = splitmix64(&x);
}
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
uint64_t orc_rng_next(orc_rng *r) {
  uint64_t *s = r->s;
  uint64_t result = rotl(s[1] * 5, 7) * 9;
  uint64_t t = s[1] << 17;
  s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
  s[2] ^= t;
  s[3] = rotl(s[3], 45);
  return result;
}
