#include "HipCiphertextFactory.hpp"

#include <random>
#include <sstream>

#include "../../include/abc_hip.h"
#include "HipCiphertext.hpp"

HipCiphertextFactory::HipCiphertextFactory() { setupContext(0); }

HipCiphertextFactory::HipCiphertextFactory(unsigned int numElementsPerCiphertextSlot, int device, uint64_t seed, size_t batchSize)
    : ciphertextSlotSize(numElementsPerCiphertextSlot), keySeed(seed), batch(batchSize) {
  if (!batch) throw std::runtime_error("HipCiphertextFactory: batch size must be at least 1");
  setupContext(device);
}

void HipCiphertextFactory::queueBatchedInput(std::vector<std::vector<int64_t>> perInstance) const {
  if (perInstance.size() != batch)
    throw std::runtime_error("queueBatchedInput: expected " + std::to_string(batch) + " vectors, got " +
                             std::to_string(perInstance.size()));
  queuedInputs.push_back(std::move(perInstance));
}

HipCiphertextFactory::~HipCiphertextFactory() {
  for (auto &e : plainCache) abc_hip_free(ctx, e.d_plain);
  abc_hip_ctx_destroy(ctx);
}

const uint64_t *HipCiphertextFactory::cachedPlaintext(const std::vector<int> &value) const {
  const std::vector<int64_t> key(value.begin(), value.end());
  for (const auto &e : plainCache)
    if (e.values == key) return e.d_plain;
  if (plainCache.size() >= kPlainCacheEntries) {
    abc_hip_free(ctx, plainCache.front().d_plain);
    plainCache.pop_front();
  }
  plainCache.push_back(CachedPlain{key, createPlaintext(key)});
  return plainCache.back().d_plain;
}

void HipCiphertextFactory::setupContext(int device) {
  // same parameter choice as SealCiphertextFactory::setupSealContext (SealCiphertextFactory.cpp:72-100)
  uint64_t primes[16];
  const int count = abc_hip_default_bfv_primes(ciphertextSlotSize, primes);
  if (count < 2) throw std::runtime_error(std::string("BFVDefault: ") + abc_hip_last_error());
  const uint64_t t = abc_hip_plain_modulus_batching(ciphertextSlotSize, 20);
  if (!t) throw std::runtime_error(std::string("PlainModulus::Batching: ") + abc_hip_last_error());
  int logn = 0;
  while ((1u << logn) < ciphertextSlotSize) ++logn;
  abcHipCheck(abc_hip_ctx_create(ABC_HIP_SCHEME_BFV, logn, primes, count, t, device, &ctx), "context");
  limbs = count - 1;
  if (!keySeed) {  // fresh keys per factory, like seal::KeyGenerator
    std::random_device rd;
    keySeed = ((uint64_t)rd() << 32) ^ rd();
  }
  abcHipCheck(abc_hip_keygen(ctx, keySeed), "key generation");
}

template <typename T>
std::vector<T> HipCiphertextFactory::expandVector(const std::vector<T> &values) const {
  std::vector<T> expanded(values.begin(), values.end());
  if (expanded.size() > ciphertextSlotSize)
    throw std::runtime_error("Cannot encode " + std::to_string(expanded.size()) + " elements in a ciphertext of size " +
                             std::to_string(ciphertextSlotSize) + ". ");
  // fill up with the last given element (SealCiphertextFactory.cpp:112-114)
  const T last = expanded.back();
  expanded.insert(expanded.end(), ciphertextSlotSize - expanded.size(), last);
  return expanded;
}

void HipCiphertextFactory::freeDevice(void *p) const {
  if (p) abc_hip_free(ctx, p);
}

uint64_t *HipCiphertextFactory::createPlaintext(const std::vector<int64_t> &value) const {
  const auto slots = expandVector(value);
  const size_t bytes = (size_t)ciphertextSlotSize * 8;
  void *d_vals = nullptr, *d_plain = nullptr;
  abcHipCheck(abc_hip_malloc(ctx, &d_vals, bytes), "plaintext allocation");
  abcHipCheck(abc_hip_malloc(ctx, &d_plain, bytes), "plaintext allocation");
  abcHipCheck(abc_hip_memcpy_h2d(ctx, d_vals, slots.data(), bytes), "plaintext upload");
  const int rc = abc_hip_batch_encode(ctx, static_cast<const int64_t *>(d_vals), static_cast<uint64_t *>(d_plain), 1);
  abc_hip_free(ctx, d_vals);
  if (rc) { abc_hip_free(ctx, d_plain); abcHipCheck(rc, "batch encode"); }
  return static_cast<uint64_t *>(d_plain);
}
uint64_t *HipCiphertextFactory::createPlaintext(const std::vector<int> &value) const {
  return createPlaintext(std::vector<int64_t>(value.begin(), value.end()));
}
uint64_t *HipCiphertextFactory::createPlaintext(int64_t value) const { return createPlaintext(std::vector<int64_t>{value}); }

std::unique_ptr<AbstractCiphertext> HipCiphertextFactory::createCiphertext(const std::vector<int64_t> &data) const {
  // B rows of N slots: the queued per-instance vectors if there are any, else B copies of `data`
  std::vector<int64_t> slots;
  slots.reserve(batch * ciphertextSlotSize);
  if (!queuedInputs.empty()) {
    for (const auto &row : queuedInputs.front()) {
      const auto e = expandVector(row);
      slots.insert(slots.end(), e.begin(), e.end());
    }
    queuedInputs.pop_front();
  } else {
    const auto e = expandVector(data);
    for (size_t b = 0; b < batch; ++b) slots.insert(slots.end(), e.begin(), e.end());
  }
  const size_t bytes = slots.size() * 8;
  void *d_vals = nullptr, *d_plain = nullptr;
  abcHipCheck(abc_hip_malloc(ctx, &d_vals, bytes), "plaintext allocation");
  if (abc_hip_malloc(ctx, &d_plain, bytes)) { abc_hip_free(ctx, d_vals); abcHipCheck(1, "plaintext allocation"); }
  int rc = abc_hip_memcpy_h2d(ctx, d_vals, slots.data(), bytes);
  if (!rc) rc = abc_hip_batch_encode(ctx, static_cast<const int64_t *>(d_vals), static_cast<uint64_t *>(d_plain), batch);
  abc_hip_free(ctx, d_vals);
  if (rc) { abc_hip_free(ctx, d_plain); abcHipCheck(rc, "batch encode"); }
  auto ctxt = std::make_unique<HipCiphertext>(std::cref(*this));
  static uint64_t encryptionCounter = 0;  // distinct randomness per encryption (instance i uses seed + i)
  encryptionCounter += batch;
  rc = abc_hip_encrypt(ctx, static_cast<const uint64_t *>(d_plain), keySeed * 0x9E3779B97F4A7C15ull + encryptionCounter,
                       ctxt->devicePtr(), batch);
  abc_hip_free(ctx, d_plain);
  abcHipCheck(rc, "encrypt");
  return ctxt;
}
std::unique_ptr<AbstractCiphertext> HipCiphertextFactory::createCiphertext(const std::vector<int> &data) const {
  return createCiphertext(std::vector<int64_t>(data.begin(), data.end()));
}
std::unique_ptr<AbstractCiphertext> HipCiphertextFactory::createCiphertext(int64_t data) const {
  return createCiphertext(std::vector<int64_t>{data});
}
std::unique_ptr<AbstractCiphertext> HipCiphertextFactory::createCiphertext(std::unique_ptr<AbstractValue> &&abstractValue) const {
  if (auto ints = dynamic_cast<Cleartext<int> *>(abstractValue.get())) {
    const auto &v = ints->getData();
    return createCiphertext(std::vector<int64_t>(v.begin(), v.end()));
  }
  throw std::runtime_error(
      "Cannot create ciphertext from any other than a Cleartext<int> as used ciphertext factory (HipCiphertextFactory) uses BFV "
      "that only supports integers.");
}

void HipCiphertextFactory::decryptCiphertextBatch(AbstractCiphertext &abstractCiphertext,
                                                  std::vector<std::vector<int64_t>> &out) const {
  auto &ctxt = dynamic_cast<HipCiphertext &>(abstractCiphertext);
  const size_t words = batch * ciphertextSlotSize, bytes = words * 8;
  void *d_plain = nullptr, *d_vals = nullptr;
  abcHipCheck(abc_hip_malloc(ctx, &d_plain, bytes), "decrypt allocation");
  if (abc_hip_malloc(ctx, &d_vals, bytes)) { abc_hip_free(ctx, d_plain); abcHipCheck(1, "decrypt allocation"); }
  int rc = abc_hip_decrypt(ctx, static_cast<const HipCiphertext &>(ctxt).devicePtr(), 2, limbs, static_cast<uint64_t *>(d_plain),
                           batch);  // read-only access: does not un-share a copy-on-write buffer
  if (!rc) rc = abc_hip_batch_decode(ctx, static_cast<const uint64_t *>(d_plain), static_cast<int64_t *>(d_vals), batch);
  std::vector<int64_t> flat(words);
  if (!rc) rc = abc_hip_memcpy_d2h(ctx, flat.data(), d_vals, bytes);  // synchronises: result observable on return
  abc_hip_free(ctx, d_plain);
  abc_hip_free(ctx, d_vals);
  abcHipCheck(rc, "decrypt");
  out.assign(batch, {});
  for (size_t b = 0; b < batch; ++b)
    out[b].assign(flat.begin() + b * ciphertextSlotSize, flat.begin() + (b + 1) * ciphertextSlotSize);
}

void HipCiphertextFactory::decryptCiphertext(AbstractCiphertext &abstractCiphertext, std::vector<int64_t> &ciphertextData) const {
  std::vector<std::vector<int64_t>> all;
  decryptCiphertextBatch(abstractCiphertext, all);
  ciphertextData = std::move(all[0]);
}

std::string HipCiphertextFactory::getString(AbstractCiphertext &abstractCiphertext) const {
  std::vector<int64_t> values;
  decryptCiphertext(abstractCiphertext, values);
  // same rendering as SealCiphertextFactory::getString (:174-189), including the seekp(-1) quirk that
  // overwrites the trailing space rather than the comma: "[ 11,  22,  33,  44, ]"
  std::stringstream ss;
  ss << "[";
  for (const auto v : values) ss << " " << v << ", ";
  ss.seekp(-1, ss.cur);
  ss << " ]";
  return ss.str();
}
