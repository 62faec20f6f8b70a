// abc_keys.hip -- key generation, encryption and decryption on the device.
//
// Reference call sites replaced (src/runtime/SealCiphertextFactory.cpp):
//   :89-93  seal::KeyGenerator: secret_key(), create_public_key, create_galois_keys (all default
//           elements), create_relin_keys                                       -> keygen
//   :12     seal::Encryptor::encrypt (public key, then modulus switch to the data level) -> encrypt
//   :150    seal::Decryptor::decrypt                                           -> decrypt
// Randomness is sampled on the host with this repo's sampling spec (DESIGN.md "Sampling spec":
// splitmix64-seeded xoshiro256**, ternary secrets, 21-vs-21-bit centred binomial errors, rejection
// sampled uniform residues) -- SEAL's own PRNG stream is not reproducible by design -- and only the
// small polynomials travel to the device; all ring arithmetic (NTTs, products, modulus switching)
// runs in HIP kernels.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <sys/random.h>
#include <thread>
#include <vector>

#include "abc_context.hpp"
#include "abc_host_math.hpp"

namespace abc {

int launch_bfv_decrypt_round(abc_hip_ctx *c, const u64 *phase, u64 *plain, size_t count);

static inline unsigned grid_for(size_t items, int block) {
  size_t g = (items + block - 1) / block;
  const size_t cap = 256 * 8 * 4;
  return (unsigned)(g < cap ? (g ? g : 1) : cap);
}

// ---------------- host samplers ----------------
// Two generators behind one interface.  Rng (splitmix64-seeded xoshiro256**) is this repo's SAMPLING SPEC for parity tests:
// the oracle implements the same stream, so keys and ciphertexts are bit-comparable -- it is NOT a cryptographic generator
// (64-bit seed, linear state) and is only reached through the explicitly seeded entry points.  ChaCha (ChaCha20 keyed with
// 256 bits from getrandom(2)) serves abc_hip_keygen_secure / abc_hip_encrypt_secure, which is what the plugin classes use
// when no test seed is given: secret material (secret key, errors, encryption randomness) and public material (the uniform
// `a` polynomials that are published inside the keys) come from independently keyed streams.
struct Sampler {
  virtual uint64_t next() = 0;
  virtual ~Sampler() {}
  int8_t ternary() {
    for (;;) {
      const uint64_t x = next();
      if (x != ~0ull) return (int8_t)((int)(x % 3) - 1);
    }
  }
  int8_t cbd() {
    const uint64_t x = next();
    return (int8_t)(__builtin_popcountll(x & 0x1FFFFF) - __builtin_popcountll((x >> 21) & 0x1FFFFF));
  }
  uint64_t uniform(uint64_t q) {
    const uint64_t lim = ~0ull - (~0ull % q) - 1;
    uint64_t x;
    do x = next(); while (x >= lim);
    return x % q;
  }
};
struct Rng final : Sampler {
  uint64_t s[4];
  explicit Rng(uint64_t seed) {
    uint64_t x = seed;
    for (auto &w : s) {
      uint64_t z = (x += 0x9E3779B97F4A7C15ull);
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      w = z ^ (z >> 31);
    }
  }
  static uint64_t rotl(uint64_t v, int k) { return (v << k) | (v >> (64 - k)); }
  uint64_t next() override {
    const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl(s[3], 45);
    return r;
  }
};
// ChaCha20 (RFC 8439 block function) as a deterministic random bit generator: 256-bit key from the operating system,
// 64-bit block counter, 64-bit stream id
struct ChaCha final : Sampler {
  uint32_t key[8], buf[16];
  uint64_t counter = 0, stream;
  int pos = 16;
  bool ok = false;
  explicit ChaCha(uint64_t stream_id) : stream(stream_id) {
    size_t got = 0;
    unsigned char *k = reinterpret_cast<unsigned char *>(key);
    while (got < sizeof(key)) {
      const ssize_t r = getrandom(k + got, sizeof(key) - got, 0);
      if (r <= 0) return;
      got += (size_t)r;
    }
    ok = true;
  }
  ~ChaCha() override {
    explicit_bzero(key, sizeof(key));
    explicit_bzero(buf, sizeof(buf));
  }
  static uint32_t rotl32(uint32_t v, int k) { return (v << k) | (v >> (32 - k)); }
  static void qr(uint32_t &a, uint32_t &b, uint32_t &c, uint32_t &d) {
    a += b; d ^= a; d = rotl32(d, 16);
    c += d; b ^= c; b = rotl32(b, 12);
    a += b; d ^= a; d = rotl32(d, 8);
    c += d; b ^= c; b = rotl32(b, 7);
  }
  void refill() {
    uint32_t st[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
    for (int i = 0; i < 8; i++) st[4 + i] = key[i];
    st[12] = (uint32_t)counter; st[13] = (uint32_t)(counter >> 32);
    st[14] = (uint32_t)stream; st[15] = (uint32_t)(stream >> 32);
    uint32_t x[16];
    for (int i = 0; i < 16; i++) x[i] = st[i];
    for (int r = 0; r < 10; r++) {
      qr(x[0], x[4], x[8], x[12]); qr(x[1], x[5], x[9], x[13]); qr(x[2], x[6], x[10], x[14]); qr(x[3], x[7], x[11], x[15]);
      qr(x[0], x[5], x[10], x[15]); qr(x[1], x[6], x[11], x[12]); qr(x[2], x[7], x[8], x[13]); qr(x[3], x[4], x[9], x[14]);
    }
    for (int i = 0; i < 16; i++) buf[i] = x[i] + st[i];
    counter++;
    pos = 0;
  }
  uint64_t next() override {
    if (pos >= 16) refill();
    const uint64_t v = (uint64_t)buf[pos] | ((uint64_t)buf[pos + 1] << 32);
    pos += 2;
    return v;
  }
};

// ---------------- kernels ----------------
// small signed polynomial [polys][N] (int8) -> residues [polys][nlm][N] for the mapped moduli
// polynomial p sits at small[(p / per) * group_stride + offset + (p % per) * N]
__global__ __launch_bounds__(256) void k_small_to_rns(DevCtx c, const int8_t *small, size_t per, size_t group_stride, size_t offset,
                                                      u64 *out, LimbMap map, int nlm, size_t polys) {
  const size_t items = polys * nlm * (size_t)c.n;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const size_t limb = it >> c.logn, x = it & (c.n - 1);
    const size_t p = limb / nlm;
    const u64 q = c.mods[map.id[limb % nlm]].q;
    const int v = small[(p / per) * group_stride + offset + (p % per) * c.n + x];
    out[it] = v < 0 ? q - (u64)(-v) : (u64)v;
  }
}

// out[polys][nlm][N] = a * b (b broadcast with stride b_stride words per poly)
__global__ __launch_bounds__(256) void k_dyadic_mul(DevCtx c, const u64 *a, const u64 *b, size_t b_stride, u64 *out, LimbMap map,
                                                    int nlm, size_t polys) {
  const size_t pw = (size_t)nlm * c.n;
  const size_t items = polys * pw;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const size_t p = it / pw, w = it % pw;
    out[it] = mul_mod(a[it], b[p * b_stride + w], c.mods[map.id[w >> c.logn]]);
  }
}

// acc[polys][nlm][N] += x   (x per-poly stride x_stride)
__global__ __launch_bounds__(256) void k_acc_add(DevCtx c, u64 *acc, const u64 *x, size_t x_stride, LimbMap map, int nlm,
                                                 size_t polys) {
  const size_t pw = (size_t)nlm * c.n;
  const size_t items = polys * pw;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const size_t p = it / pw, w = it % pw;
    acc[it] = add_mod(acc[it], x[p * x_stride + w], c.mods[map.id[w >> c.logn]].q);
  }
}

// key-switch key assembly (KeyGenerator::generate_one_kswitch_key):
//   key[i][0][j] = -(a_i[j]*s[j] + e_i[j]) (+ (q_sp mod q_i) * new_key[i] when j == i),  key[i][1][j] = a_i[j]
// a: [L][K][N] uniform (NTT domain), e: [L][K][N] error already in NTT form, s/new_key: [K][N] NTT form
__global__ __launch_bounds__(256) void k_make_kskey(DevCtx c, const u64 *a, const u64 *e, const u64 *s, const u64 *new_key,
                                                    u64 *key, int nkeys) {
  const size_t pw = (size_t)c.K * c.n;
  const size_t items = (size_t)nkeys * pw;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const int i = (int)(it / pw);
    const size_t w = it % pw;
    const int j = (int)(w >> c.logn);
    const size_t x = w & (c.n - 1);
    const Mod m = c.mods[j];
    const u64 av = a[it];
    u64 v = neg_mod(add_mod(mul_mod(av, s[w], m), e[it], m.q), m.q);
    if (new_key && j == i) v = add_mod(v, mul_mod(new_key[(size_t)i * c.n + x], c.cst->special_mod_q[i], m), m.q);
    key[((size_t)i * 2 + 0) * pw + w] = v;
    key[((size_t)i * 2 + 1) * pw + w] = av;
  }
}

// prodS/prodD split of [polys][K][N] into data limbs [polys][L][N] and the special limb [polys][N]
static int split_special(abc_hip_ctx *c, const u64 *full, u64 *data, u64 *special, size_t polys) {
  const size_t N = (size_t)c->n;
  ABC_HIP_CHECK(hipMemcpy2DAsync(data, c->L * N * 8, full, c->K * N * 8, c->L * N * 8, polys, hipMemcpyDeviceToDevice, c->stream));
  ABC_HIP_CHECK(hipMemcpy2DAsync(special, N * 8, full + (size_t)c->L * N, c->K * N * 8, N * 8, polys, hipMemcpyDeviceToDevice,
                                 c->stream));
  return 0;
}

// c[ct][p][j] = u[ct][j] * pk[p][j]   (key level, NTT form)
__global__ __launch_bounds__(256) void k_enc_mul_pk(DevCtx c, const u64 *u, const u64 *pk, u64 *cfull, size_t count) {
  const size_t pw = (size_t)c.K * c.n;
  const size_t items = count * 2 * pw;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const size_t ct = it / (2 * pw), r = it % (2 * pw);
    const size_t w = r % pw;
    cfull[it] = mul_mod(u[ct * pw + w], pk[r], c.mods[w >> c.logn]);
  }
}

// ---------------- key generation ----------------
// pub draws the uniform `a` polynomials (published in the key), sec the errors; the seeded spec passes one generator as both
// (P, S: the generators' concrete -- final -- types, so the 156 M draws of a key set at N = 2^16 inline and the rejection limit of
// `uniform` is computed once per prime instead of once per draw)
template <class P, class S>
static int make_kskey(abc_hip_ctx *c, P &pub, S &sec, const u64 *d_new_key, u64 *d_key, std::vector<uint64_t> &h_a,
                      std::vector<int8_t> &h_e, u64 *d_a, int8_t *d_e8, u64 *d_e, int nkeys) {
  const size_t N = (size_t)c->n;
  const int K = c->K;
  for (int i = 0; i < nkeys; i++) {
    for (int j = 0; j < K; j++)
      for (size_t x = 0; x < N; x++) h_a[((size_t)i * K + j) * N + x] = pub.uniform(c->primes[j]);
    for (size_t x = 0; x < N; x++) h_e[(size_t)i * N + x] = sec.cbd();
  }
  ABC_HIP_CHECK(hipMemcpyAsync(d_a, h_a.data(), (size_t)nkeys * K * N * 8, hipMemcpyHostToDevice, c->stream));
  ABC_HIP_CHECK(hipMemcpyAsync(d_e8, h_e.data(), (size_t)nkeys * N, hipMemcpyHostToDevice, c->stream));
  LimbMap kmap{};
  for (int j = 0; j < K; j++) kmap.id[j] = j;
  hipLaunchKernelGGL(k_small_to_rns, dim3(grid_for((size_t)nkeys * K * N, 256)), dim3(256), 0, c->stream, c->dc, d_e8, (size_t)1, N,
                     (size_t)0, d_e, kmap, K, (size_t)nkeys);
  ABC_HIP_CHECK(hipGetLastError());
  if (launch_ntt_fwd(c, d_e, kmap, K, (size_t)nkeys * K)) return 1;
  hipLaunchKernelGGL(k_make_kskey, dim3(grid_for((size_t)nkeys * K * N, 256)), dim3(256), 0, c->stream, c->dc, d_a, d_e, c->d_sk,
                     d_new_key, d_key, nkeys);
  ABC_HIP_CHECK(hipGetLastError());
  ABC_HIP_CHECK(hipStreamSynchronize(c->stream));  // host staging buffers are reused by the next key
  return 0;
}

template <class P, class S>
static int keygen_with(abc_hip_ctx *c, P &pub, S &sec);
int keygen(abc_hip_ctx *c, uint64_t seed) {
  Rng rng(seed);
  return keygen_with(c, rng, rng);
}
int keygen_secure(abc_hip_ctx *c) {
  ChaCha sec(1), pub(2);  // independently keyed: nothing derived from the secret stream is ever published
  if (!sec.ok || !pub.ok) { set_error("keygen: getrandom failed"); return 1; }
  return keygen_with(c, pub, sec);
}
template <class P, class S>
static int keygen_with(abc_hip_ctx *c, P &pub, S &sec) {
  const size_t N = (size_t)c->n;
  const int K = c->K, L = c->L;
  LimbMap kmap{};
  for (int j = 0; j < K; j++) kmap.id[j] = j;
  // staging
  std::vector<uint64_t> h_a((size_t)L * K * N);
  std::vector<int8_t> h_e((size_t)L * N);
  u64 *d_a = nullptr, *d_e = nullptr, *d_newkey = nullptr;
  int8_t *d_e8 = nullptr;
  ABC_HIP_CHECK(hipMalloc(&d_a, (size_t)L * K * N * 8));
  ABC_HIP_CHECK(hipMalloc(&d_e, (size_t)L * K * N * 8));
  ABC_HIP_CHECK(hipMalloc(&d_e8, (size_t)L * N));
  ABC_HIP_CHECK(hipMalloc(&d_newkey, (size_t)K * N * 8));
  // secret key
  for (size_t x = 0; x < N; x++) h_e[x] = sec.ternary();
  ABC_HIP_CHECK(hipMemcpyAsync(d_e8, h_e.data(), N, hipMemcpyHostToDevice, c->stream));
  if (!c->d_sk) ABC_HIP_CHECK(hipMalloc(&c->d_sk, (size_t)K * N * 8));
  hipLaunchKernelGGL(k_small_to_rns, dim3(grid_for((size_t)K * N, 256)), dim3(256), 0, c->stream, c->dc, d_e8, (size_t)1, N, (size_t)0,
                     c->d_sk, kmap, K, (size_t)1);
  ABC_HIP_CHECK(hipGetLastError());
  if (launch_ntt_fwd(c, c->d_sk, kmap, K, K)) return 1;
  ABC_HIP_CHECK(hipStreamSynchronize(c->stream));
  // public key = one symmetric encryption of zero at key level
  if (!c->d_pk) ABC_HIP_CHECK(hipMalloc(&c->d_pk, (size_t)2 * K * N * 8));
  if (make_kskey(c, pub, sec, nullptr, c->d_pk, h_a, h_e, d_a, d_e8, d_e, 1)) return 1;
  // relinearisation key: switches s^2 -> s
  drop_key_twins(c, nullptr);  // every key-switching key is about to be regenerated
  if (!c->d_relin) ABC_HIP_CHECK(hipMalloc(&c->d_relin, c->key_words() * 8));
  hipLaunchKernelGGL(k_dyadic_mul, dim3(grid_for((size_t)K * N, 256)), dim3(256), 0, c->stream, c->dc, c->d_sk, c->d_sk,
                     (size_t)0, d_newkey, kmap, K, (size_t)1);
  ABC_HIP_CHECK(hipGetLastError());
  if (make_kskey(c, pub, sec, d_newkey, c->d_relin, h_a, h_e, d_a, d_e8, d_e, L)) return 1;
  // Galois keys for the default element set (GaloisTool::get_elts_all): 2N-1, then 3^(2^i), 3^-(2^i)
  for (auto &kv : c->d_galois) (void)hipFree(kv.second);
  c->d_galois.clear();
  c->galois_order.clear();
  const uint64_t m = 2 * (uint64_t)N;
  std::vector<uint32_t> elts;
  elts.push_back((uint32_t)(m - 1));
  uint64_t pos = 3, neg = host::invmod(3, m);
  for (int i = 0; i < c->logn - 1; i++) {
    elts.push_back((uint32_t)pos); pos = (pos * pos) & (m - 1);
    elts.push_back((uint32_t)neg); neg = (neg * neg) & (m - 1);
  }
  for (uint32_t elt : elts) {
    if (c->d_galois.count(elt)) continue;  // 3^(N/4) = 3^-(N/4) mod 2N is listed twice: first key wins
    u64 *d_key = nullptr;
    ABC_HIP_CHECK(hipMalloc(&d_key, c->key_words() * 8));
    if (launch_galois(c, c->d_sk, d_newkey, K, 1, elt, true)) return 1;
    if (make_kskey(c, pub, sec, d_newkey, d_key, h_a, h_e, d_a, d_e8, d_e, L)) return 1;
    c->d_galois[elt] = d_key;
    c->galois_order.push_back(elt);
  }
  // the staging buffers held the secret key, s^2 / g(s) and the errors: wipe before release
  (void)hipMemsetAsync(d_e, 0, (size_t)L * K * N * 8, c->stream);
  (void)hipMemsetAsync(d_e8, 0, (size_t)L * N, c->stream);
  (void)hipMemsetAsync(d_newkey, 0, (size_t)K * N * 8, c->stream);
  ABC_HIP_CHECK(hipStreamSynchronize(c->stream));
  explicit_bzero(h_e.data(), h_e.size());
  (void)hipFree(d_a); (void)hipFree(d_e); (void)hipFree(d_e8); (void)hipFree(d_newkey);
  return 0;
}

// ---------------- encryption ----------------
static int encrypt_with(abc_hip_ctx *c, const u64 *plain, uint64_t seed, Sampler *secure, u64 *ct, size_t count);
int encrypt(abc_hip_ctx *c, const u64 *plain, uint64_t seed, u64 *ct, size_t count) {
  return encrypt_with(c, plain, seed, nullptr, ct, count);
}
int encrypt_secure(abc_hip_ctx *c, const u64 *plain, u64 *ct, size_t count) {
  ChaCha sec(3);  // a fresh 256-bit key from the operating system per call: nothing is derived from the key seed
  if (!sec.ok) { set_error("encrypt: getrandom failed"); return 1; }
  return encrypt_with(c, plain, 0, &sec, ct, count);
}
static int encrypt_with(abc_hip_ctx *c, const u64 *plain, uint64_t seed, Sampler *secure, u64 *ct, size_t count) {
  if (!c->d_pk) { set_error("encrypt: no public key (call abc_hip_keygen or abc_hip_load_public_key)"); return 1; }
  if (!count) return 0;
  const size_t N = (size_t)c->n;
  const int K = c->K, L = c->L;
  const bool ckks = (c->scheme == 2);
  // host sampling: per ciphertext i the stream seed+i yields u, e0, e1 -- independent streams, so a batch is drawn by up to 16
  // host threads (the seeded spec gives the same bytes whatever the thread count; the secure path keys one ChaCha20 instance per
  // worker from the operating system).  Config 5 encrypts 1 125 ciphertexts of 2^16 slots: 221 M draws, 530 ms on one thread.
  std::vector<int8_t> h_small(count * 3 * N);
  {
    auto fill = [N](auto &rng, int8_t *p) {  // concrete (final) generator type: the draws inline
      for (size_t x = 0; x < N; x++) p[x] = rng.ternary();
      for (size_t x = 0; x < 2 * N; x++) p[N + x] = rng.cbd();
    };
    const size_t hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    const size_t workers = count >= 4 ? std::min(hw, count) : 1;
    std::atomic<int> failed{0};
    auto work = [&](size_t w) {
      const size_t i0 = count * w / workers, i1 = count * (w + 1) / workers;
      if (secure) {
        if (w == 0) {  // the caller's generator serves the first share
          for (size_t i = i0; i < i1; i++) {
            int8_t *p = h_small.data() + i * 3 * N;
            for (size_t x = 0; x < N; x++) p[x] = secure->ternary();
            for (size_t x = 0; x < 2 * N; x++) p[N + x] = secure->cbd();
          }
          return;
        }
        ChaCha own(0x100 + w);
        if (!own.ok) { failed = 1; return; }
        for (size_t i = i0; i < i1; i++) fill(own, h_small.data() + i * 3 * N);
        return;
      }
      for (size_t i = i0; i < i1; i++) {
        Rng seeded(seed + i);
        fill(seeded, h_small.data() + i * 3 * N);
      }
    };
    if (workers == 1) {
      work(0);
    } else {
      std::vector<std::thread> pool;
      for (size_t w = 1; w < workers; w++) pool.emplace_back(work, w);
      work(0);
      for (auto &t : pool) t.join();
    }
    if (failed) { explicit_bzero(h_small.data(), h_small.size()); set_error("encrypt: getrandom failed"); return 1; }
  }
  // workspace: small 3N bytes | u [K][N] | cfull [2][K][N] | err [2][K][N] | prodD [2][L][N] | prodS [2][N] | tmod [2][L][N]
  const size_t per_ct_words = (size_t)(K + 2 * K + 2 * K + 2 * L + 2 + 2 * L) * N;
  const size_t small_bytes = (count * 3 * N + 7) / 8 * 8;
  if (ensure_workspace(c, small_bytes + count * per_ct_words * 8)) return 1;
  int8_t *d_small = (int8_t *)c->ws;
  u64 *u = (u64 *)((char *)c->ws + small_bytes);
  u64 *cfull = u + count * K * N, *err = cfull + count * 2 * K * N;
  u64 *prodD = err + count * 2 * K * N, *prodS = prodD + count * 2 * L * N, *tmod = prodS + count * 2 * N;
  ABC_HIP_CHECK(hipMemcpyAsync(d_small, h_small.data(), count * 3 * N, hipMemcpyHostToDevice, c->stream));
  ABC_HIP_CHECK(hipStreamSynchronize(c->stream));  // h_small is a local
  explicit_bzero(h_small.data(), h_small.size());
  LimbMap kmap{};
  for (int j = 0; j < K; j++) kmap.id[j] = j;
  // u (poly 0 of each [3][N] triple) -> residues at key level -> NTT
  hipLaunchKernelGGL(k_small_to_rns, dim3(grid_for(count * K * N, 256)), dim3(256), 0, c->stream, c->dc, d_small, (size_t)1, 3 * N,
                     (size_t)0, u, kmap, K, count);
  ABC_HIP_CHECK(hipGetLastError());
  if (launch_ntt_fwd(c, u, kmap, K, count * K)) return 1;
  // errors e0,e1 (polys 1,2 of each triple) -> residues [count][2][K][N]
  hipLaunchKernelGGL(k_small_to_rns, dim3(grid_for(count * 2 * K * N, 256)), dim3(256), 0, c->stream, c->dc, d_small, (size_t)2,
                     3 * N, N, err, kmap, K, count * 2);
  ABC_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(k_enc_mul_pk, dim3(grid_for(count * 2 * K * N, 256)), dim3(256), 0, c->stream, c->dc, u, c->d_pk, cfull, count);
  ABC_HIP_CHECK(hipGetLastError());
  if (ckks) {
    if (launch_ntt_fwd(c, err, kmap, K, count * 2 * K)) return 1;
  } else {
    if (launch_ntt_inv(c, cfull, kmap, K, count * 2 * K)) return 1;
  }
  hipLaunchKernelGGL(k_acc_add, dim3(grid_for(count * 2 * K * N, 256)), dim3(256), 0, c->stream, c->dc, cfull, err,
                     (size_t)K * N, kmap, K, count * 2);
  ABC_HIP_CHECK(hipGetLastError());
  // modulus switch key level -> data level (divide_and_round_q_last(_ntt)_inplace)
  if (split_special(c, cfull, prodD, prodS, count * 2)) return 1;
  LimbMap smap{};
  smap.id[0] = K - 1;
  if (ckks && launch_ntt_inv(c, prodS, smap, 1, count * 2)) return 1;
  if (launch_ks_tmod(c, prodS, tmod, L, count * 2)) return 1;
  const LimbMap dmap = key_limb_map(c, L);
  if (ckks && launch_ntt_fwd(c, tmod, dmap, L, count * 2 * L)) return 1;
  if (launch_ks_finish(c, prodD, tmod, ct, nullptr, 0, false, L, count)) return 1;
  // the encryption randomness (u, e0, e1 and everything derived before the modulus switch) lives in the workspace: wipe it
  if (secure) ABC_HIP_CHECK(hipMemsetAsync(c->ws, 0, small_bytes + count * (size_t)(5 * K) * N * 8, c->stream));
  // add the message
  if (ckks) return ckks_add_plain(c, ct, plain, (size_t)L * N, ct, 2, L, count, 0);
  return bfv_addsub_plain(c, ct, plain, N, ct, 2, count, 0);
}

// ---------------- decryption ----------------
// phase = c0 + c1*s (+ c2*s^2) at the ciphertext's level
__global__ __launch_bounds__(256) void k_phase_mul(DevCtx c, const u64 *cn, const u64 *sk, u64 *acc, int size, int nl, size_t count) {
  // cn: [count][size-1][nl][N] NTT form of c1.. ; acc [count][nl][N]
  const size_t pw = (size_t)nl * c.n;
  const size_t items = count * pw;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += stride) {
    const size_t ct = it / pw, w = it % pw;
    const Mod m = c.mods[w >> c.logn];
    const u64 s = sk[w];
    u64 sp = s, sum = 0;
    for (int p = 1; p < size; p++) {
      sum = add_mod(sum, mul_mod(cn[(ct * (size - 1) + (p - 1)) * pw + w], sp, m), m.q);
      if (p + 1 < size) sp = mul_mod(sp, s, m);
    }
    acc[it] = sum;
  }
}

int decrypt(abc_hip_ctx *c, const u64 *ct, int size, int nl, u64 *plain, size_t count) {
  if (!c->d_sk) { set_error("decrypt: no secret key"); return 1; }
  if (size < 2 || size > 3) { set_error("decrypt: ciphertext size must be 2 or 3"); return 1; }
  if (!count) return 0;
  const size_t N = (size_t)c->n;
  const bool ckks = (c->scheme == 2);
  if (!ckks && nl != c->L) { set_error("decrypt: BFV ciphertexts live at the top level"); return 1; }
  const size_t pw = (size_t)nl * N;
  if (ensure_workspace(c, (count * (size - 1) * pw + count * pw) * 8)) return 1;
  u64 *cn = (u64 *)c->ws, *acc = cn + count * (size - 1) * pw;
  const LimbMap dmap = key_limb_map(c, nl);
  // copy c1.. (strided inside each ciphertext)
  ABC_HIP_CHECK(hipMemcpy2DAsync(cn, (size - 1) * pw * 8, ct + pw, size * pw * 8, (size - 1) * pw * 8, count,
                                 hipMemcpyDeviceToDevice, c->stream));
  if (!ckks && launch_ntt_fwd(c, cn, dmap, nl, count * (size - 1) * nl)) return 1;
  hipLaunchKernelGGL(k_phase_mul, dim3(grid_for(count * pw, 256)), dim3(256), 0, c->stream, c->dc, cn, c->d_sk, acc, size, nl,
                     count);
  ABC_HIP_CHECK(hipGetLastError());
  if (!ckks && launch_ntt_inv(c, acc, dmap, nl, count * nl)) return 1;
  // + c0
  u64 *dst = ckks ? plain : acc;
  if (ckks) ABC_HIP_CHECK(hipMemcpyAsync(plain, acc, count * pw * 8, hipMemcpyDeviceToDevice, c->stream));
  hipLaunchKernelGGL(k_acc_add, dim3(grid_for(count * pw, 256)), dim3(256), 0, c->stream, c->dc, dst, ct, (size_t)size * pw, dmap,
                     nl, count);
  ABC_HIP_CHECK(hipGetLastError());
  if (ckks) return 0;
  return launch_bfv_decrypt_round(c, acc, plain, count);
}

}  // namespace abc
