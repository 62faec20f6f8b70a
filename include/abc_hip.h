/*
 * abc_hip.h -- C ABI of libabc_hip.so: the MI355X (gfx950) FHE runtime backend that replaces
 * Microsoft SEAL behind ABC's AbstractCiphertext / AbstractCiphertextFactory plugin surface.
 *
 * Every entry point names the reference interface it replaces (paths relative to the ABC tree).
 * Conventions
 *   - every function returns 0 on success, non-zero on error; abc_hip_last_error() gives the text
 *     (the C++ shim turns it into std::runtime_error, the reference's only error convention:
 *     src/runtime/SealCiphertext.cpp:40,137).
 *   - `d_` pointers are DEVICE pointers obtained from abc_hip_malloc (or any HIP allocation, e.g. a
 *     torch tensor's data_ptr); `h_` pointers are host pointers.
 *   - all residues are uint64; layouts are row-major
 *        ciphertext batch  [count][size][nl][N]   BFV: coefficient form, CKKS: NTT form
 *        BFV plaintext     [count][N]             coefficients mod t
 *        CKKS plaintext    [count][nl][N]         NTT form
 *        key-switch key    [L][2][L+1][N]         (decomposition limb, component, key-level limb), NTT form
 *     `count` independent ciphertexts are processed by one call (the batch dimension the reference's
 *     single-circuit RuntimeVisitor lacks, include/ast_opt/runtime/RuntimeVisitor.h:34).
 *   - work is enqueued on the context's stream (abc_hip_set_stream); results are observable after
 *     abc_hip_sync or any *_d2h copy.
 */
#ifndef ABC_HIP_H
#define ABC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ABC_HIP_SCHEME_BFV 1  /* seal::scheme_type::bfv, src/runtime/SealCiphertextFactory.cpp:74 */
#define ABC_HIP_SCHEME_CKKS 2 /* HAVE_SEAL_CKKS, CMakeLists.txt:216 (no reference implementation) */

typedef struct abc_hip_ctx abc_hip_ctx;

const char *abc_hip_last_error(void);
int abc_hip_device_count(void);

/* ---- context: replaces SealCiphertextFactory::setupSealContext, SealCiphertextFactory.cpp:72-100 ----
 * primes = data limbs followed by the special key-switching prime (SEAL's key-level chain);
 * plain_modulus is ignored for CKKS. */
int abc_hip_ctx_create(int scheme, int logn, const uint64_t *h_primes, int nprimes, uint64_t plain_modulus, int device,
                       abc_hip_ctx **out);
void abc_hip_ctx_destroy(abc_hip_ctx *ctx);
/* seal::CoeffModulus::BFVDefault(N) (SealCiphertextFactory.cpp:80); returns the number of primes */
int abc_hip_default_bfv_primes(size_t n, uint64_t *h_out);
/* seal::PlainModulus::Batching(N, bits) (SealCiphertextFactory.cpp:83) */
uint64_t abc_hip_plain_modulus_batching(size_t n, int bits);
/* seal::CoeffModulus::Create(N, bit_sizes) ordering, for CKKS chains */
int abc_hip_create_primes(size_t n, const int *bit_sizes, int count, uint64_t *h_out);
int abc_hip_ctx_info(const abc_hip_ctx *ctx, int what); /* 0 scheme, 1 logn, 2 nprimes, 3 L, 4 device */
/* Every operation of the context is enqueued on `hip_stream` from now on.  NULL does NOT mean HIP's legacy default
 * stream: it selects the context's own private non-blocking stream again (the state after abc_hip_ctx_create), which is
 * not ordered against anything else -- a caller that produces inputs on another stream (e.g. torch's current stream, whose
 * handle is 0 for the default stream) must synchronise that stream before the call and abc_hip_sync after it, or pass a
 * real stream handle here (bench.py creates a torch.cuda.Stream and passes its handle). */
int abc_hip_set_stream(abc_hip_ctx *ctx, void *hip_stream);
int abc_hip_sync(abc_hip_ctx *ctx);
/* The ABC_HIP_* path switches (README) are read when the context is created, not per operation; this re-reads them
 * (drains the stream first).  For A/B timing and the parity tests of the fallback paths. */
int abc_hip_ctx_reload_env(abc_hip_ctx *ctx);

/* ---- device memory (so that FFI callers need no HIP runtime binding) ----
 * Freed buffers are cached per context and recycled by size, so a buffer may be freed right after the last operation
 * on it has been issued, without synchronising; it must only be used with the context that allocated it.
 * ABC_HIP_SYNC_ALLOC=1 selects plain hipMalloc / synchronise + hipFree. */
int abc_hip_malloc(abc_hip_ctx *ctx, void **d_ptr, size_t bytes);
int abc_hip_free(abc_hip_ctx *ctx, void *d_ptr);
/* hand every cached (freed, not yet reused) buffer back to the driver; also done automatically when a device
 * allocation of this context fails and when the cache reaches its cap (a quarter of the device) */
int abc_hip_trim(abc_hip_ctx *ctx);
size_t abc_hip_cached_bytes(abc_hip_ctx *ctx);
int abc_hip_memcpy_h2d(abc_hip_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int abc_hip_memcpy_d2h(abc_hip_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
/* seal::Ciphertext copy-ctor = SealCiphertext::clone, src/runtime/SealCiphertext.cpp:13-16,71-78 */
int abc_hip_memcpy_d2d(abc_hip_ctx *ctx, void *d_dst, const void *d_src, size_t bytes);

/* ---- keys: replaces seal::KeyGenerator use at SealCiphertextFactory.cpp:89-93 ---- */
/* generate sk, pk, relin key and all default Galois keys on the device.
 * abc_hip_keygen_secure: what a deployment uses (and HipCiphertextFactory's default): secret key and errors from ChaCha20
 * keyed with 256 bits of getrandom(2), the published uniform polynomials from an independently keyed stream, secret
 * temporaries wiped before release.
 * abc_hip_keygen(seed): TEST ONLY -- the repo's reproducible sampling spec (splitmix64-seeded xoshiro256**),
 * which the CPU oracle implements too so that keys are bit-comparable; a 64-bit seed and a linear generator are not
 * cryptographic strength. */
int abc_hip_keygen_secure(abc_hip_ctx *ctx);
int abc_hip_keygen(abc_hip_ctx *ctx, uint64_t seed);
/* or load externally generated keys (host pointers) */
int abc_hip_load_secret_key(abc_hip_ctx *ctx, const uint64_t *h_sk /*[L+1][N] NTT*/);
int abc_hip_load_public_key(abc_hip_ctx *ctx, const uint64_t *h_pk /*[2][L+1][N]*/);
int abc_hip_load_relin_key(abc_hip_ctx *ctx, const uint64_t *h_key /*[L][2][L+1][N]*/);
int abc_hip_load_galois_key(abc_hip_ctx *ctx, uint32_t galois_elt, const uint64_t *h_key);
int abc_hip_get_secret_key(abc_hip_ctx *ctx, uint64_t *h_out);
int abc_hip_get_public_key(abc_hip_ctx *ctx, uint64_t *h_out);
int abc_hip_get_relin_key(abc_hip_ctx *ctx, uint64_t *h_out);
int abc_hip_get_galois_key(abc_hip_ctx *ctx, uint32_t galois_elt, uint64_t *h_out);
int abc_hip_num_galois_keys(abc_hip_ctx *ctx);
uint32_t abc_hip_galois_elt_at(abc_hip_ctx *ctx, int i);
uint32_t abc_hip_galois_elt_from_step(abc_hip_ctx *ctx, int step);

/* ---- encode / encrypt / decrypt / decode ----
 * seal::BatchEncoder::encode (SealCiphertextFactory.cpp:130): d_values int64 [count][N] (already padded
 * by the caller as SealCiphertextFactory::expandVector does, :102-115) -> d_plain [count][N] */
int abc_hip_batch_encode(abc_hip_ctx *ctx, const int64_t *d_values, uint64_t *d_plain, size_t count);
/* seal::BatchEncoder::decode (SealCiphertextFactory.cpp:151) */
int abc_hip_batch_decode(abc_hip_ctx *ctx, const uint64_t *d_plain, int64_t *d_values, size_t count);
/* seal::Encryptor::encrypt, public key (SealCiphertextFactory.cpp:12).
 * abc_hip_encrypt_secure: encryption randomness from a freshly OS-keyed ChaCha20 stream per call (never derived from a
 * key seed), wiped afterwards.  abc_hip_encrypt(seed): TEST ONLY, ciphertext i uses the reproducible stream seed+i. */
int abc_hip_encrypt_secure(abc_hip_ctx *ctx, const uint64_t *d_plain, uint64_t *d_ct, size_t count);
int abc_hip_encrypt(abc_hip_ctx *ctx, const uint64_t *d_plain, uint64_t seed, uint64_t *d_ct, size_t count);
/* seal::Decryptor::decrypt (SealCiphertextFactory.cpp:150); size = 2 or 3 polynomials */
int abc_hip_decrypt(abc_hip_ctx *ctx, const uint64_t *d_ct, int size, int nl, uint64_t *d_plain, size_t count);

/* ---- evaluator: replaces the seal::Evaluator calls in src/runtime/SealCiphertext.cpp ---- */
/* Evaluator::add / add_inplace (:92,:114) */
int abc_hip_add(abc_hip_ctx *ctx, const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out, int size, int nl, size_t count);
/* Evaluator::sub / sub_inplace (:98,:118) */
int abc_hip_sub(abc_hip_ctx *ctx, const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out, int size, int nl, size_t count);
/* Evaluator::negate / negate_inplace (:157,:193) */
int abc_hip_negate(abc_hip_ctx *ctx, const uint64_t *d_a, uint64_t *d_out, int size, int nl, size_t count);
/* Evaluator::multiply(_inplace) (:104,:122): size-2 x size-2 -> size-3, no relinearisation */
int abc_hip_multiply(abc_hip_ctx *ctx, const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out3, int nl, size_t count);
/* Evaluator::relinearize_inplace (:105,:123,:160,:197): size-3 -> size-2 */
int abc_hip_relinearize(abc_hip_ctx *ctx, const uint64_t *d_ct3, uint64_t *d_out2, int nl, size_t count);
/* SealCiphertext::multiply / multiplyInplace = multiply + relinearize (:102-107,:121-124) -- THE hot path */
int abc_hip_mul_relin(abc_hip_ctx *ctx, const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out, int nl, size_t count);
/* Evaluator::rotate_rows(_inplace) (:55,:60), incl. SEAL's NAF decomposition for steps without a key */
int abc_hip_rotate(abc_hip_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, int nl, int steps, size_t count);
/* Evaluator::apply_galois for one element */
int abc_hip_apply_galois(abc_hip_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, int nl, uint32_t galois_elt, size_t count);
/* Evaluator::multiply_plain(_inplace) (:159,:196); plain_stride = 0 broadcasts one plaintext to the batch */
int abc_hip_multiply_plain(abc_hip_ctx *ctx, const uint64_t *d_ct, const uint64_t *d_plain, size_t plain_stride,
                           uint64_t *d_out, int size, int nl, size_t count);
/* Evaluator::add_plain(_inplace) (:134,:175) */
int abc_hip_add_plain(abc_hip_ctx *ctx, const uint64_t *d_ct, const uint64_t *d_plain, size_t plain_stride, uint64_t *d_out,
                      int size, int nl, size_t count);
/* Evaluator::sub_plain(_inplace) (:145,:184) */
int abc_hip_sub_plain(abc_hip_ctx *ctx, const uint64_t *d_ct, const uint64_t *d_plain, size_t plain_stride, uint64_t *d_out,
                      int size, int nl, size_t count);
/* CKKS only: Evaluator::rescale_to_next / mod_switch_to_next (no reference call site).
 * d_in [count][size][nl][N] -> d_out [count][size][nl-1][N]; d_out must not alias d_in (error otherwise). */
int abc_hip_rescale(abc_hip_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, int size, int nl, size_t count);
int abc_hip_mod_switch(abc_hip_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, int size, int nl, size_t count);

/* ---- HIP-graph capture of an operation sequence (a recorded circuit in place of the
 * eager per-call dispatch of SpecialRuntimeVisitor, src/runtime/RuntimeVisitor.cpp:40-159).
 * Everything enqueued on the context between begin and end is recorded instead of executed; the sequence must have
 * run once eagerly before (so that no scratch allocation happens while capturing) and may only use device pointers
 * that stay valid for every launch.  abc_hip_encrypt / abc_hip_keygen / *_h2d / *_d2h are not capturable.
 * Buffers: abc_hip_malloc inside a capture is served from the cache only (never the driver).  Every buffer the recorded
 * sequence can have touched is pinned to the graph until abc_hip_graph_destroy: the blocks the capture itself allocated or
 * freed, and EVERY abc_hip_malloc block of this context that is still out with the caller when abc_hip_graph_end runs (so an
 * operand that existed before the capture and is freed only after it -- a cached plaintext, an input -- is covered too).
 * Freeing a pinned block parks it instead of recycling it, so a replay can never run over memory that has been handed to
 * someone else; a block pinned by several graphs returns to the cache when the last of them is destroyed.  Memory that did not
 * come from abc_hip_malloc is the caller's to keep alive.  A buffer that existed before the capture
 * and is freed inside it is an INPUT of the circuit: it keeps its address and contents are the caller's to refresh before a
 * replay (HipCiphertextFactory::rewriteCiphertext). */
int abc_hip_graph_begin(abc_hip_ctx *ctx);
int abc_hip_graph_end(abc_hip_ctx *ctx, void **graph_exec_out);
int abc_hip_graph_launch(abc_hip_ctx *ctx, void *graph_exec);
int abc_hip_graph_destroy(abc_hip_ctx *ctx, void *graph_exec);

/* ---- raw transforms, exposed for kernel-level parity tests and profiling ---- */
/* mod_kind: 0 = key-level prime `index`, 1 = BEHZ Bsk prime `index`, 2 = plaintext modulus */
int abc_hip_ntt_forward(abc_hip_ctx *ctx, uint64_t *d_data, int mod_kind, int index, size_t count);
int abc_hip_ntt_inverse(abc_hip_ctx *ctx, uint64_t *d_data, int mod_kind, int index, size_t count);
/* forward / inverse transform of whole polynomials at a data level: d_data [polys][nl][N], limb j modulo q_j
 * (CKKS plaintexts and ciphertexts travel in NTT form; host-side encoders produce coefficient form) */
int abc_hip_ntt_limbs(abc_hip_ctx *ctx, uint64_t *d_data, int nl, size_t polys, int inverse);
/* key-switch contribution only: d_target [count][nl][N] -> d_out2 [count][2][nl][N]; key_kind 0 relin, else Galois elt */
int abc_hip_keyswitch(abc_hip_ctx *ctx, const uint64_t *d_target, uint32_t key_kind, uint64_t *d_out2, int nl, size_t count);
/* micro-benchmarks of the integer / fp64 pipes (returns elapsed ms for `iters` dependent modmuls per lane); probe kernels,
 * present only in a library built with -DABC_HIP_WITH_MICROBENCH (python -m abc_amd.build --microbench), an error otherwise */
int abc_hip_microbench(abc_hip_ctx *ctx, int which, int iters, double *ms_out);
/* elapsed milliseconds between two HIP events recorded on the context stream */
int abc_hip_timer_start(abc_hip_ctx *ctx);
int abc_hip_timer_stop(abc_hip_ctx *ctx, float *ms_out);

#ifdef __cplusplus
}
#endif
#endif
