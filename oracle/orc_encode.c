/*
 * oracle/orc_encode.c -- TEST INFRASTRUCTURE (CPU oracle), see orc_internal.h header.
 *
 * CKKS canonical-embedding encoder/decoder (double precision, host side).  The reference has no
 * CKKS code at all (src/runtime/SealCiphertextFactory.cpp:74 hard-codes bfv), so this
 * follows the published CKKS encoding: slot i <-> evaluation at zeta^(3^i), zeta = exp(i*pi/N), conjugate
 * slots at zeta^(-3^i) [SEAL-recall: CKKSEncoder, generator 3].  Floating point: NOT a bit-parity
 * target; decoded values are compared with a tolerance in the tests.
 */
#include "orc_internal.h"
#include "oracle.h"

#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static void fft_inplace(double complex *a, size_t n, int sign) {
  /* iterative radix-2, sign=-1: forward e^{-2 pi i jk/n}, sign=+1: backward (unnormalised) */
  int logn = 0;
  while (((size_t)1 << logn) < n) logn++;
  for (size_t i = 0; i < n; i++) {
    size_t j = orc_bitrev((uint32_t)i, logn);
    if (j > i) { double complex t = a[i]; a[i] = a[j]; a[j] = t; }
  }
  for (size_t len = 2; len <= n; len <<= 1) {
    double ang = sign * 2.0 * M_PI / (double)len;
    for (size_t i = 0; i < n; i += len)
      for (size_t k = 0; k < len / 2; k++) {
        double complex w = cos(ang * (double)k) + I * sin(ang * (double)k);
        double complex u = a[i + k], v = a[i + k + len / 2] * w;
        a[i + k] = u + v;
        a[i + k + len / 2] = u - v;
      }
  }
}

int orc_ckks_encode(const orc_ctx *c, const double *re, const double *im, size_t count, double scale, int nl,
                    uint64_t *plain) {
  size_t n = c->n, slots = n >> 1, m2 = n << 1;
  if (count > slots) return -1;
  double complex *w = (double complex *)calloc(n, sizeof(double complex));
  uint64_t g = 1;
  for (size_t i = 0; i < slots; i++) {
    double complex z = i < count ? re[i] + I * (im ? im[i] : 0.0) : 0.0;
    w[(g - 1) >> 1] = z;
    w[(m2 - g - 1) >> 1] = conj(z);
    g = (g * 3) & (m2 - 1);
  }
  fft_inplace(w, n, -1);
  int rc = 0;
  for (size_t k = 0; k < n; k++) {
    double ang = -M_PI * (double)k / (double)n;
    double complex v = w[k] * (cos(ang) + I * sin(ang)) / (double)n;
    double coef = creal(v) * scale;
    if (fabs(coef) >= 9.0e18) { rc = -2; break; }
    long long r = llround(coef);
    for (int j = 0; j < nl; j++) {
      uint64_t q = c->qmod[j].q;
      uint64_t mag = (uint64_t)(r < 0 ? -r : r) % q;
      plain[(size_t)j * n + k] = r < 0 ? orc_neg_mod(mag, q) : mag;
    }
  }
  free(w);
  if (rc) return rc;
  for (int j = 0; j < nl; j++) orc_ntt_fwd(plain + (size_t)j * n, &c->ntt[j]);
  return 0;
}

/* little multiword helpers (words little-endian, fixed length W) */
#define W (ORC_MAX_LIMBS + 1)
static void mw_mul_add(uint64_t *x, uint64_t mul, uint64_t add) { /* x = x*mul + add */
  u128 carry = add;
  for (int i = 0; i < W; i++) {
    u128 p = (u128)x[i] * mul + carry;
    x[i] = (uint64_t)p;
    carry = p >> 64;
  }
}
static int mw_cmp(const uint64_t *a, const uint64_t *b) {
  for (int i = W - 1; i >= 0; i--) if (a[i] != b[i]) return a[i] > b[i] ? 1 : -1;
  return 0;
}
static void mw_sub(uint64_t *r, const uint64_t *a, const uint64_t *b) { /* r = a - b, a >= b */
  uint64_t borrow = 0;
  for (int i = 0; i < W; i++) {
    uint64_t bi = b[i] + borrow;
    uint64_t nb = (bi < borrow) || (a[i] < bi);
    r[i] = a[i] - bi;
    borrow = nb;
  }
}
static long double mw_to_ld(const uint64_t *a) {
  long double v = 0.0L;
  for (int i = W - 1; i >= 0; i--) v = v * 18446744073709551616.0L + (long double)a[i];
  return v;
}

int orc_ckks_decode(const orc_ctx *c, const uint64_t *plain, int nl, double scale, double *re, double *im) {
  size_t n = c->n, slots = n >> 1, m2 = n << 1;
  uint64_t *coef = (uint64_t *)malloc((size_t)nl * n * 8);
  memcpy(coef, plain, (size_t)nl * n * 8);
  for (int j = 0; j < nl; j++) orc_ntt_inv(coef + (size_t)j * n, &c->ntt[j]);
  /* q and q/2 as multiword */
  uint64_t Q[W], Qh[W];
  memset(Q, 0, sizeof(Q));
  Q[0] = 1;
  for (int j = 0; j < nl; j++) mw_mul_add(Q, c->qmod[j].q, 0);
  for (int i = 0; i < W; i++) Qh[i] = (Q[i] >> 1) | (i + 1 < W ? Q[i + 1] << 63 : 0);
  /* Garner constants: inv of prod_{i<j} q_i mod q_j */
  uint64_t inv_rad[ORC_MAX_LIMBS];
  for (int j = 0; j < nl; j++) {
    uint64_t qj = c->qmod[j].q, rad = 1 % qj;
    for (int i = 0; i < j; i++) rad = orc_mul_mod(rad, c->qmod[i].q % qj, qj);
    inv_rad[j] = orc_inv_mod(rad, qj);
  }
  double complex *w = (double complex *)malloc(n * sizeof(double complex));
  for (size_t k = 0; k < n; k++) {
    uint64_t d[ORC_MAX_LIMBS];
    for (int j = 0; j < nl; j++) {
      uint64_t qj = c->qmod[j].q;
      uint64_t acc = 0, rad = 1 % qj;
      for (int i = 0; i < j; i++) {
        acc = orc_add_mod(acc, orc_mul_mod(d[i] % qj, rad, qj), qj);
        rad = orc_mul_mod(rad, c->qmod[i].q % qj, qj);
      }
      d[j] = orc_mul_mod(orc_sub_mod(coef[(size_t)j * n + k], acc, qj), inv_rad[j], qj);
    }
    uint64_t x[W];
    memset(x, 0, sizeof(x));
    for (int j = nl - 1; j >= 0; j--) mw_mul_add(x, c->qmod[j].q, d[j]); /* Horner over mixed radix */
    long double val;
    if (mw_cmp(x, Qh) > 0) { uint64_t y[W]; mw_sub(y, Q, x); val = -mw_to_ld(y); }
    else val = mw_to_ld(x);
    double ang = M_PI * (double)k / (double)n;
    w[k] = (double)(val / (long double)scale) * (cos(ang) + I * sin(ang));
  }
  fft_inplace(w, n, +1);
  uint64_t g = 1;
  for (size_t i = 0; i < slots; i++) {
    double complex z = w[(g - 1) >> 1];
    re[i] = creal(z);
    if (im) im[i] = cimag(z);
    g = (g * 3) & (m2 - 1);
  }
  free(w); free(coef);
  return 0;
}
