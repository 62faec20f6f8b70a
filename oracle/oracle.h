/*
 * oracle/oracle.h -- public (ctypes-friendly) API of the CPU oracle.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE: a CPU restatement of the SEAL 3.6 algorithms the
 * reference's runtime calls (src/runtime/SealCiphertext.cpp, src/runtime/SealCiphertextFactory.cpp).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so.
 * Parity: decrypted-slot level pinned by the reference's golden vectors (tests/test_oracle_golden.py);
 * ciphertext-residue level vs. a real SEAL binary: parity unpinned (SEAL absent from /root/reference).
 *
 * Layouts (uint64, row-major):
 *   plaintext (BFV)  [N]            coefficients mod t
 *   plaintext (CKKS) [nl][N]        NTT form
 *   ciphertext       [size][nl][N]  BFV: coefficient form; CKKS: NTT form (bit-reversed order)
 *   key-switch key   [L][2][L+1][N] (decomposition limb, component, key-level limb), NTT form
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ctx orc_ctx;

/* parameters */
int orc_default_bfv_primes(size_t n, uint64_t *out);                       /* returns count */
int orc_create_primes(size_t n, const int *bit_sizes, int count, uint64_t *out);
uint64_t orc_plain_modulus_batching(size_t n, int bits);
int orc_is_prime(uint64_t v);

/* context */
orc_ctx *orc_ctx_create(int scheme, int logn, const uint64_t *primes, int nprimes, uint64_t plain_modulus);
void orc_ctx_destroy(orc_ctx *c);
int orc_ctx_info(const orc_ctx *c, int what); /* 0 scheme,1 logn,2 nkey,3 L,4 nBsk,5 ngalois */
uint64_t orc_ctx_prime(const orc_ctx *c, int i);
uint64_t orc_ctx_plain_modulus(const orc_ctx *c);
uint64_t orc_ctx_ntt_root(const orc_ctx *c, int i);
uint64_t orc_ctx_behz_prime(const orc_ctx *c, int which);

/* raw transforms (kernel-level parity) */
void orc_ntt_forward(const orc_ctx *c, int prime_index, uint64_t *a);
void orc_ntt_inverse(const orc_ctx *c, int prime_index, uint64_t *a);
void orc_ntt_forward_behz(const orc_ctx *c, int bsk_index, uint64_t *a);
void orc_ntt_inverse_behz(const orc_ctx *c, int bsk_index, uint64_t *a);
void orc_ntt_forward_plain(const orc_ctx *c, uint64_t *a);
void orc_ntt_inverse_plain(const orc_ctx *c, uint64_t *a);

/* keys */
int orc_keygen(orc_ctx *c, uint64_t seed);
int orc_get_secret_key(const orc_ctx *c, uint64_t *out);   /* [L+1][N] NTT form */
int orc_get_public_key(const orc_ctx *c, uint64_t *out);   /* [2][L+1][N] */
int orc_get_relin_key(const orc_ctx *c, uint64_t *out);    /* [L][2][L+1][N] */
int orc_num_galois(const orc_ctx *c);
uint32_t orc_galois_elt_at(const orc_ctx *c, int i);
int orc_get_galois_key(const orc_ctx *c, uint32_t elt, uint64_t *out);
uint32_t orc_galois_elt_from_step(const orc_ctx *c, int step);
int orc_naf(int value, int *out);                          /* returns count */

/* BFV encode / encrypt / decrypt */
int orc_batch_encode(const orc_ctx *c, const int64_t *values, size_t count, uint64_t *plain);
int orc_batch_decode(const orc_ctx *c, const uint64_t *plain, int64_t *values);
int orc_bfv_encrypt(const orc_ctx *c, const uint64_t *plain, uint64_t seed, uint64_t *ct);
int orc_bfv_decrypt(const orc_ctx *c, const uint64_t *ct, int size, uint64_t *plain);
int orc_bfv_noise_budget(const orc_ctx *c, const uint64_t *ct, int size);

/* evaluator -- exact residue semantics of seal::Evaluator */
int orc_add(const orc_ctx *c, const uint64_t *a, const uint64_t *b, int size, int nl, uint64_t *out);
int orc_sub(const orc_ctx *c, const uint64_t *a, const uint64_t *b, int size, int nl, uint64_t *out);
int orc_negate(const orc_ctx *c, const uint64_t *a, int size, int nl, uint64_t *out);
int orc_bfv_multiply(const orc_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out3);
int orc_relinearize(const orc_ctx *c, const uint64_t *ct3, int nl, uint64_t *out2);
int orc_bfv_mul_relin(const orc_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out2);
int orc_rotate(const orc_ctx *c, const uint64_t *ct, int nl, int steps, uint64_t *out2);
int orc_apply_galois(const orc_ctx *c, const uint64_t *ct, int nl, uint32_t elt, uint64_t *out2);
int orc_bfv_multiply_plain(const orc_ctx *c, const uint64_t *ct, int size, const uint64_t *plain, uint64_t *out);
int orc_bfv_add_plain(const orc_ctx *c, const uint64_t *ct, int size, const uint64_t *plain, uint64_t *out);
int orc_bfv_sub_plain(const orc_ctx *c, const uint64_t *ct, int size, const uint64_t *plain, uint64_t *out);

/* CKKS (no reference implementation exists, SURVEY.md section 0: semantics follow SEAL's CKKS evaluator) */
int orc_ckks_multiply(const orc_ctx *c, const uint64_t *a, const uint64_t *b, int nl, uint64_t *out3);
int orc_ckks_mul_relin(const orc_ctx *c, const uint64_t *a, const uint64_t *b, int nl, uint64_t *out2);
int orc_ckks_rescale(const orc_ctx *c, const uint64_t *ct, int size, int nl, uint64_t *out);
int orc_ckks_mod_switch(const orc_ctx *c, const uint64_t *ct, int size, int nl, uint64_t *out);
int orc_ckks_multiply_plain(const orc_ctx *c, const uint64_t *ct, int size, int nl, const uint64_t *plain, uint64_t *out);
int orc_ckks_add_plain(const orc_ctx *c, const uint64_t *ct, int size, int nl, const uint64_t *plain, uint64_t *out);
int orc_ckks_encode(const orc_ctx *c, const double *re, const double *im, size_t count, double scale, int nl,
                    uint64_t *plain);
int orc_ckks_decode(const orc_ctx *c, const uint64_t *plain, int nl, double scale, double *re, double *im);
int orc_ckks_encrypt(const orc_ctx *c, const uint64_t *plain, uint64_t seed, uint64_t *ct);
int orc_ckks_decrypt(const orc_ctx *c, const uint64_t *ct, int size, int nl, uint64_t *plain);

/* stand-alone pieces exposed for kernel-level parity tests */
int orc_keyswitch(const orc_ctx *c, const uint64_t *target, int nl, const uint64_t *key, uint64_t *out2);
int orc_galois_permute(const orc_ctx *c, const uint64_t *poly, int nl, uint32_t elt, int ntt_form, uint64_t *out);

/* CPU-baseline timing loop: runs `iters` mul+relin on the same operands, returns seconds */
double orc_time_mul_relin(const orc_ctx *c, const uint64_t *a, const uint64_t *b, int nl, int iters, uint64_t *out2);

#ifdef __cplusplus
}
#endif
#endif
