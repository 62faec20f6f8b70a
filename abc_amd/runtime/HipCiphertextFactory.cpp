#include "HipCiphertextFactory.hpp"

#include <cmath>
#include <sstream>

#include "../../include/abc_hip.h"
#include "HipCiphertext.hpp"
#include "SealWire.hpp"

HipCiphertextFactory::HipCiphertextFactory() { setupContext(0); }

HipCiphertextFactory::HipCiphertextFactory(unsigned int numElementsPerCiphertextSlot, int device, uint64_t seed, size_t batchSize)
    : ciphertextSlotSize(numElementsPerCiphertextSlot), keySeed(seed), batch(batchSize) {
  if (!batch) throw std::runtime_error("HipCiphertextFactory: batch size must be at least 1");
  setupContext(device);
}

HipCiphertextFactory::HipCiphertextFactory(const HipSchemeConfig &cfg)
    : ciphertextSlotSize(cfg.ringDegree), keySeed(cfg.seed), batch(cfg.batch), ckksMode(cfg.ckks), ckksScale(cfg.ckksScale),
      ckksBits(cfg.ckksBits) {
  if (!batch) throw std::runtime_error("HipCiphertextFactory: batch size must be at least 1");
  if (ckksMode && ckksBits.size() < 2) throw std::runtime_error("HipCiphertextFactory: a CKKS chain needs a data limb and the special prime");
  setupContext(cfg.device);
}

void HipCiphertextFactory::queueBatchedRealInput(std::vector<std::vector<double>> perInstance) const {
  if (perInstance.size() != batch)
    throw std::runtime_error("queueBatchedRealInput: expected " + std::to_string(batch) + " vectors, got " +
                             std::to_string(perInstance.size()));
  queuedRealInputs.push_back(std::move(perInstance));
}

void HipCiphertextFactory::queueBatchedInput(std::vector<std::vector<int64_t>> perInstance) const {
  if (perInstance.size() != batch)
    throw std::runtime_error("queueBatchedInput: expected " + std::to_string(batch) + " vectors, got " +
                             std::to_string(perInstance.size()));
  queuedInputs.push_back(std::move(perInstance));
}

HipCiphertextFactory::~HipCiphertextFactory() {
  for (auto &e : plainCache) abc_hip_free(ctx, e.d_plain);
  for (auto &e : ckksPlainCache) abc_hip_free(ctx, e.d_plain);
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
  uint64_t primes[16];
  int count = 0;
  uint64_t t = 0;
  int logn = 0;
  while ((1u << logn) < ciphertextSlotSize) ++logn;
  if (ckksMode) {
    // seal::CoeffModulus::Create(N, bit_sizes): what a HAVE_SEAL_CKKS build of the reference would pass to its context
    count = (int)ckksBits.size();
    if (count > 16) throw std::runtime_error("HipCiphertextFactory: at most 16 primes");
    abcHipCheck(abc_hip_create_primes(ciphertextSlotSize, ckksBits.data(), count, primes), "CoeffModulus::Create");
  } else {
    // same parameter choice as SealCiphertextFactory::setupSealContext (SealCiphertextFactory.cpp:72-100)
    count = abc_hip_default_bfv_primes(ciphertextSlotSize, primes);
    if (count < 2) throw std::runtime_error(std::string("BFVDefault: ") + abc_hip_last_error());
    t = abc_hip_plain_modulus_batching(ciphertextSlotSize, 20);
    if (!t) throw std::runtime_error(std::string("PlainModulus::Batching: ") + abc_hip_last_error());
  }
  abcHipCheck(abc_hip_ctx_create(ckksMode ? ABC_HIP_SCHEME_CKKS : ABC_HIP_SCHEME_BFV, logn, primes, count, t, device, &ctx), "context");
  limbs = count - 1;
  chain.assign(primes, primes + count);
  plainModulus = t;
  if (ckksMode) ckksEncoder = CkksEncoder(ciphertextSlotSize, std::vector<uint64_t>(primes, primes + limbs));
  // fresh keys per factory, like seal::KeyGenerator: OS-keyed ChaCha20 unless a TEST seed asks for the reproducible spec
  if (keySeed) abcHipCheck(abc_hip_keygen(ctx, keySeed), "key generation");
  else abcHipCheck(abc_hip_keygen_secure(ctx), "key generation");
}

template <typename T>
std::vector<T> HipCiphertextFactory::expandVector(const std::vector<T> &values) const {
  std::vector<T> expanded(values.begin(), values.end());
  const size_t slots = usableSlots();
  if (expanded.size() > slots)
    throw std::runtime_error("Cannot encode " + std::to_string(expanded.size()) + " elements in a ciphertext of size " +
                             std::to_string(slots) + ". ");
  if (expanded.empty()) throw std::runtime_error("Cannot encode an empty vector.");
  // fill up with the last given element (SealCiphertextFactory.cpp:112-114)
  const T last = expanded.back();
  expanded.insert(expanded.end(), slots - expanded.size(), last);
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
  if (ckksMode) {
    if (!queuedInputs.empty()) {  // integer batches queued for a CKKS factory: serve them as reals
      std::vector<std::vector<double>> rows;
      for (const auto &row : queuedInputs.front()) rows.emplace_back(row.begin(), row.end());
      queuedInputs.pop_front();
      queuedRealInputs.push_front(std::move(rows));
    }
    return createCkksCiphertext(std::vector<double>(data.begin(), data.end()));
  }
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
  rc = encryptInto(d_plain, ctxt->devicePtr());
  abc_hip_free(ctx, d_plain);
  abcHipCheck(rc, "encrypt");
  return ctxt;
}

int HipCiphertextFactory::encryptInto(const void *d_plain, uint64_t *d_ct) const {
  if (!keySeed) return abc_hip_encrypt_secure(ctx, static_cast<const uint64_t *>(d_plain), d_ct, batch);
  // TEST seed given: reproducible randomness, distinct per encryption (instance i of this call uses seed + i)
  static uint64_t encryptionCounter = 0;
  encryptionCounter += batch;
  return abc_hip_encrypt(ctx, static_cast<const uint64_t *>(d_plain), keySeed * 0x9E3779B97F4A7C15ull + encryptionCounter, d_ct, batch);
}

std::unique_ptr<AbstractCiphertext> HipCiphertextFactory::createCiphertext(const std::vector<double> &data) const {
  if (!ckksMode) throw std::runtime_error("Cannot create a ciphertext from real values: this factory uses BFV that only supports integers.");
  return createCkksCiphertext(data);
}

std::unique_ptr<AbstractCiphertext> HipCiphertextFactory::createCkksCiphertext(const std::vector<double> &data) const {
  // B plaintexts [L][N]: encode on the host (floating point, off the hot path), forward transform + encryption on the device
  const size_t N = ciphertextSlotSize, per = (size_t)limbs * N;
  std::vector<uint64_t> coeffs(batch * per);
  if (!queuedRealInputs.empty()) {
    for (size_t b = 0; b < batch; ++b) ckksEncoder.encode(expandVector(queuedRealInputs.front()[b]), ckksScale, limbs, coeffs.data() + b * per);
    queuedRealInputs.pop_front();
  } else {
    ckksEncoder.encode(expandVector(data), ckksScale, limbs, coeffs.data());
    for (size_t b = 1; b < batch; ++b) std::copy(coeffs.begin(), coeffs.begin() + per, coeffs.begin() + b * per);
  }
  void *d_plain = nullptr;
  abcHipCheck(abc_hip_malloc(ctx, &d_plain, coeffs.size() * 8), "plaintext allocation");
  int rc = abc_hip_memcpy_h2d(ctx, d_plain, coeffs.data(), coeffs.size() * 8);
  if (!rc) rc = abc_hip_ntt_limbs(ctx, static_cast<uint64_t *>(d_plain), limbs, batch, 0);
  if (rc) { abc_hip_free(ctx, d_plain); abcHipCheck(rc, "CKKS encode"); }
  auto ctxt = std::make_unique<HipCiphertext>(std::cref(*this));  // top level, default scale
  rc = encryptInto(d_plain, ctxt->devicePtr());
  abc_hip_free(ctx, d_plain);
  abcHipCheck(rc, "encrypt");
  return ctxt;
}

const uint64_t *HipCiphertextFactory::cachedCkksPlaintext(const std::vector<double> &value, int level, double scale) const {
  for (const auto &e : ckksPlainCache)
    if (e.level == level && e.scale == scale && e.values == value) return e.d_plain;
  if (ckksPlainCache.size() >= kPlainCacheEntries) {
    abc_hip_free(ctx, ckksPlainCache.front().d_plain);
    ckksPlainCache.pop_front();
  }
  const size_t N = ciphertextSlotSize;
  std::vector<uint64_t> coeffs((size_t)level * N);
  ckksEncoder.encode(expandVector(value), scale, level, coeffs.data());
  void *d_plain = nullptr;
  abcHipCheck(abc_hip_malloc(ctx, &d_plain, coeffs.size() * 8), "plaintext allocation");
  int rc = abc_hip_memcpy_h2d(ctx, d_plain, coeffs.data(), coeffs.size() * 8);
  if (!rc) rc = abc_hip_ntt_limbs(ctx, static_cast<uint64_t *>(d_plain), level, 1, 0);
  if (rc) { abc_hip_free(ctx, d_plain); abcHipCheck(rc, "CKKS encode"); }
  ckksPlainCache.push_back(CachedCkksPlain{value, level, scale, static_cast<uint64_t *>(d_plain)});
  return ckksPlainCache.back().d_plain;
}

void HipCiphertextFactory::decryptCiphertextRealBatch(AbstractCiphertext &abstractCiphertext, std::vector<std::vector<double>> &out) const {
  if (!ckksMode) throw std::runtime_error("decryptCiphertextReal: not a CKKS factory");
  const auto &ctxt = dynamic_cast<const HipCiphertext &>(abstractCiphertext);
  const int nl = ctxt.level();
  const size_t N = ciphertextSlotSize, per = (size_t)nl * N, bytes = batch * per * 8;
  void *d_plain = nullptr;
  abcHipCheck(abc_hip_malloc(ctx, &d_plain, bytes), "decrypt allocation");
  int rc = abc_hip_decrypt(ctx, ctxt.devicePtr(), 2, nl, static_cast<uint64_t *>(d_plain), batch);
  if (!rc) rc = abc_hip_ntt_limbs(ctx, static_cast<uint64_t *>(d_plain), nl, batch, 1);
  std::vector<uint64_t> coeffs(batch * per);
  if (!rc) rc = abc_hip_memcpy_d2h(ctx, coeffs.data(), d_plain, bytes);  // synchronises: result observable on return
  abc_hip_free(ctx, d_plain);
  abcHipCheck(rc, "decrypt");
  out.assign(batch, {});
  for (size_t b = 0; b < batch; ++b) ckksEncoder.decode(coeffs.data() + b * per, nl, ctxt.scale(), out[b]);
}
void HipCiphertextFactory::decryptCiphertextReal(AbstractCiphertext &abstractCiphertext, std::vector<double> &out) const {
  std::vector<std::vector<double>> all;
  decryptCiphertextRealBatch(abstractCiphertext, all);
  out = std::move(all[0]);
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
  if (ckksMode) {
    if (auto d = dynamic_cast<Cleartext<double> *>(abstractValue.get())) return createCkksCiphertext(d->getData());
    if (auto fl = dynamic_cast<Cleartext<float> *>(abstractValue.get()))
      return createCkksCiphertext(std::vector<double>(fl->getData().begin(), fl->getData().end()));
  }
  throw std::runtime_error(
      "Cannot create ciphertext from any other than a Cleartext<int> as used ciphertext factory (HipCiphertextFactory) uses BFV "
      "that only supports integers.");
}

void HipCiphertextFactory::decryptCiphertextBatch(AbstractCiphertext &abstractCiphertext,
                                                  std::vector<std::vector<int64_t>> &out) const {
  if (ckksMode) {  // the interface speaks int64: nearest integers of the decoded slots
    std::vector<std::vector<double>> real;
    decryptCiphertextRealBatch(abstractCiphertext, real);
    out.assign(real.size(), {});
    for (size_t b = 0; b < real.size(); ++b)
      for (double v : real[b]) out[b].push_back((int64_t)std::llrint(v));
    return;
  }
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

// ---- GraphCapable ----
void HipCiphertextFactory::graphBegin() const { abcHipCheck(abc_hip_graph_begin(ctx), "graph capture"); }
void *HipCiphertextFactory::graphEnd() const {
  void *g = nullptr;
  abcHipCheck(abc_hip_graph_end(ctx, &g), "graph capture");
  return g;
}
void HipCiphertextFactory::graphAbort() const {
  void *g = nullptr;
  if (abc_hip_graph_end(ctx, &g) == 0 && g) abc_hip_graph_destroy(ctx, g);
}
void HipCiphertextFactory::graphLaunch(void *graph) const { abcHipCheck(abc_hip_graph_launch(ctx, graph), "graph launch"); }
void HipCiphertextFactory::graphDestroy(void *graph) const { abcHipCheck(abc_hip_graph_destroy(ctx, graph), "graph destroy"); }
void HipCiphertextFactory::synchronize() const { abcHipCheck(abc_hip_sync(ctx), "synchronize"); }

void HipCiphertextFactory::rewriteCiphertextBatch(AbstractCiphertext &target, const std::vector<std::vector<int64_t>> &perInstance) const {
  if (ckksMode) queueBatchedRealInput([&] {
    std::vector<std::vector<double>> rows;
    for (const auto &r : perInstance) rows.emplace_back(r.begin(), r.end());
    return rows;
  }());
  else queueBatchedInput(perInstance);
  rewriteCiphertext(target, perInstance.at(0));
}
void HipCiphertextFactory::rewriteCiphertext(AbstractCiphertext &target, const std::vector<int64_t> &values) const {
  auto &dst = dynamic_cast<HipCiphertext &>(target);
  if (dst.level() != limbs) throw std::runtime_error("rewriteCiphertext: the target must be a fresh (top-level) ciphertext");
  auto fresh = createCiphertext(values);  // serves a queued batch if there is one
  // straight into the existing buffer, shared or not: every holder of this buffer is meant to see the new input
  abcHipCheck(abc_hip_memcpy_d2d(ctx, const_cast<uint64_t *>(static_cast<const HipCiphertext &>(dst).devicePtr()),
                                 dynamic_cast<const HipCiphertext &>(*fresh).devicePtr(), ciphertextWords() * 8),
              "rewrite ciphertext");
}

// ---- SEAL 3.6 wire format ----
namespace {
sealwire::Parms wireParms(bool ckks, unsigned int n, const std::vector<uint64_t> &chain, uint64_t t) {
  sealwire::Parms p;
  p.scheme = ckks ? 2 : 1;
  p.ringDegree = n;
  p.primes = chain;
  p.plainModulus = ckks ? 0 : t;
  return p;
}
sealwire::Compression wireMode(int compression) {
  if (compression < 0 || compression > 2) throw std::runtime_error("SEAL wire format: compression must be 0 (none), 1 (zlib) or 2 (zstd)");
  return (sealwire::Compression)compression;
}
}  // namespace

namespace {
sealwire::CiphertextImage keyImage(const sealwire::ParmsId &id, size_t N, size_t K, const uint64_t *words) {
  sealwire::CiphertextImage img;
  img.id = id;
  img.nttForm = true;
  img.size = 2;
  img.ringDegree = N;
  img.limbs = K;
  img.scale = 1.0;
  img.data.assign(words, words + 2 * K * N);
  return img;
}
// residues must be canonical: the device kernels take words below their prime (the fp64 ones convert them exactly only below 2^52)
void checkReduced(const std::vector<uint64_t> &data, size_t polys, size_t N, const std::vector<uint64_t> &primes, const char *what) {
  const size_t K = primes.size();
  for (size_t p = 0; p < polys; ++p)
    for (size_t j = 0; j < K; ++j) {
      const uint64_t q = primes[j];
      const uint64_t *w = data.data() + (p * K + j) * N;
      for (size_t i = 0; i < N; ++i)
        if (w[i] >= q) throw std::runtime_error(std::string(what) + ": residue not reduced modulo its prime");
    }
}
void checkKeyImage(const sealwire::CiphertextImage &img, const sealwire::ParmsId &id, size_t N, const std::vector<uint64_t> &primes,
                   const char *what) {
  const size_t K = primes.size();
  if (img.id != id) throw std::runtime_error(std::string(what) + ": parms_id is not this factory's key-level parameter set");
  if (!img.nttForm || img.size != 2 || img.ringDegree != N || img.limbs != K)
    throw std::runtime_error(std::string(what) + ": expected a size-2 NTT-form key over all key primes");
  checkReduced(img.data, 2, N, primes, what);
}
}  // namespace

void HipCiphertextFactory::saveCiphertext(const AbstractCiphertext &ciphertext, std::ostream &out, int compression) const {
  const auto &c = dynamic_cast<const HipCiphertext &>(ciphertext);
  if (&c.getFactory() != this) throw std::runtime_error("saveCiphertext: ciphertext belongs to another factory");
  const size_t N = ciphertextSlotSize, per = (size_t)2 * c.level() * N;
  std::vector<uint64_t> host(batch * per);
  abcHipCheck(abc_hip_memcpy_d2h(ctx, host.data(), c.devicePtr(), host.size() * 8), "ciphertext download");
  sealwire::CiphertextImage img;
  // a ciphertext names the parameter set of ITS level: the first `level` data primes
  img.id = sealwire::parmsId(wireParms(ckksMode, ciphertextSlotSize, chain, plainModulus), (size_t)c.level());
  img.nttForm = ckksMode;  // SEAL keeps BFV ciphertexts in coefficient form and CKKS ones in NTT form; so does this backend
  img.size = 2;
  img.ringDegree = N;
  img.limbs = (uint64_t)c.level();
  img.scale = ckksMode ? c.scale() : 1.0;
  for (size_t b = 0; b < batch; ++b) {
    img.data.assign(host.begin() + b * per, host.begin() + (b + 1) * per);
    sealwire::save(out, img, wireMode(compression));
  }
}

std::unique_ptr<AbstractCiphertext> HipCiphertextFactory::loadCiphertext(std::istream &in) const {
  const size_t N = ciphertextSlotSize;
  const sealwire::Parms parms = wireParms(ckksMode, ciphertextSlotSize, chain, plainModulus);
  std::vector<uint64_t> host;
  int level = 0;
  double scale = 1.0;
  for (size_t b = 0; b < batch; ++b) {
    sealwire::CiphertextImage img;
    sealwire::load(in, img, (uint64_t)3 * chain.size() * N);
    int found = 0;
    for (int l = limbs; l >= 1 && !found; --l)
      if (img.id == sealwire::parmsId(parms, (size_t)l)) found = l;
    if (!found) throw std::runtime_error("loadCiphertext: parms_id matches no level of this factory's parameters (scheme, N, primes, t)");
    if (img.size != 2) throw std::runtime_error("loadCiphertext: only relinearised (size 2) ciphertexts travel through the plugin surface");
    if (img.ringDegree != N || img.limbs != (uint64_t)found) throw std::runtime_error("loadCiphertext: dimensions contradict parms_id");
    if (img.nttForm != ckksMode) throw std::runtime_error("loadCiphertext: unexpected representation (BFV: coefficient form, CKKS: NTT form)");
    if (ckksMode) {  // the scale comes off the wire: finite, positive and below the modulus of its level, or every later check is moot
      double bits = 0.0;
      for (int j = 0; j < found; ++j) bits += std::log2((double)chain[j]);
      if (!std::isfinite(img.scale) || !(img.scale > 0.0) || !(std::log2(img.scale) < bits))
        throw std::runtime_error("loadCiphertext: scale out of bounds for the ciphertext's level");
    }
    if (b == 0) {
      level = found;
      scale = ckksMode ? img.scale : 1.0;
      host.reserve(batch * img.data.size());
    } else if (found != level || (ckksMode && img.scale != scale)) {
      throw std::runtime_error("loadCiphertext: the instances of one batched value must share level and scale");
    }
    for (size_t j = 0; j < (size_t)found; ++j)
      for (int comp = 0; comp < 2; ++comp)
        for (size_t i = 0; i < N; ++i)
          if (img.data[((size_t)comp * found + j) * N + i] >= chain[j]) throw std::runtime_error("loadCiphertext: residue not reduced modulo its prime");
    host.insert(host.end(), img.data.begin(), img.data.end());
  }
  auto c = std::make_unique<HipCiphertext>(std::cref(*this), level, scale);
  abcHipCheck(abc_hip_memcpy_h2d(ctx, c->devicePtr(), host.data(), host.size() * 8), "ciphertext upload");
  return c;
}

void HipCiphertextFactory::saveSecretKey(std::ostream &out, int compression) const {
  const size_t N = ciphertextSlotSize, K = chain.size();
  sealwire::PlaintextImage img;
  img.id = sealwire::parmsId(wireParms(ckksMode, ciphertextSlotSize, chain, plainModulus), K);
  img.coeffCount = K * N;
  img.scale = 1.0;
  img.data.resize(K * N);
  abcHipCheck(abc_hip_get_secret_key(ctx, img.data.data()), "secret key download");
  sealwire::save(out, img, wireMode(compression));
  std::fill(img.data.begin(), img.data.end(), 0);
}
void HipCiphertextFactory::loadSecretKey(std::istream &in) {
  const size_t N = ciphertextSlotSize, K = chain.size();
  sealwire::PlaintextImage img;
  sealwire::load(in, img, (uint64_t)K * N);
  if (img.id != sealwire::parmsId(wireParms(ckksMode, ciphertextSlotSize, chain, plainModulus), K))
    throw std::runtime_error("loadSecretKey: parms_id is not this factory's key-level parameter set");
  if (img.coeffCount != K * N) throw std::runtime_error("loadSecretKey: expected N coefficients per key prime (NTT form)");
  checkReduced(img.data, 1, N, chain, "loadSecretKey");
  const int rc = abc_hip_load_secret_key(ctx, img.data.data());
  std::fill(img.data.begin(), img.data.end(), 0);
  abcHipCheck(rc, "secret key upload");
}


void HipCiphertextFactory::savePublicKey(std::ostream &out, int compression) const {
  const size_t N = ciphertextSlotSize, K = chain.size();
  std::vector<uint64_t> words(2 * K * N);
  abcHipCheck(abc_hip_get_public_key(ctx, words.data()), "public key download");
  const auto id = sealwire::parmsId(wireParms(ckksMode, ciphertextSlotSize, chain, plainModulus), K);
  sealwire::save(out, keyImage(id, N, K, words.data()), wireMode(compression));
}
void HipCiphertextFactory::loadPublicKey(std::istream &in) {
  const size_t N = ciphertextSlotSize, K = chain.size();
  sealwire::CiphertextImage img;
  sealwire::load(in, img, (uint64_t)2 * K * N);
  checkKeyImage(img, sealwire::parmsId(wireParms(ckksMode, ciphertextSlotSize, chain, plainModulus), K), N, chain, "loadPublicKey");
  abcHipCheck(abc_hip_load_public_key(ctx, img.data.data()), "public key upload");
}

void HipCiphertextFactory::saveRelinKeys(std::ostream &out, int compression) const {
  const size_t N = ciphertextSlotSize, K = chain.size(), per = 2 * K * N;
  std::vector<uint64_t> words((size_t)limbs * per);
  abcHipCheck(abc_hip_get_relin_key(ctx, words.data()), "relinearisation key download");
  sealwire::KSwitchImage img;
  img.id = sealwire::parmsId(wireParms(ckksMode, ciphertextSlotSize, chain, plainModulus), K);
  img.keys.resize(1);  // seal::RelinKeys: entry 0 <-> the key for s^2
  for (int j = 0; j < limbs; ++j) img.keys[0].push_back(keyImage(img.id, N, K, words.data() + (size_t)j * per));
  sealwire::save(out, img, wireMode(compression));
}
void HipCiphertextFactory::loadRelinKeys(std::istream &in) {
  const size_t N = ciphertextSlotSize, K = chain.size(), per = 2 * K * N;
  sealwire::KSwitchImage img;
  sealwire::load(in, img, (uint64_t)(limbs + 1) * per);
  const auto id = sealwire::parmsId(wireParms(ckksMode, ciphertextSlotSize, chain, plainModulus), K);
  if (img.id != id) throw std::runtime_error("loadRelinKeys: parms_id is not this factory's key-level parameter set");
  if (img.keys.empty() || img.keys[0].size() != (size_t)limbs)
    throw std::runtime_error("loadRelinKeys: expected one entry (s^2) with one key per data prime");
  std::vector<uint64_t> words;
  words.reserve((size_t)limbs * per);
  for (const auto &pk : img.keys[0]) {
    checkKeyImage(pk, id, N, chain, "loadRelinKeys");
    words.insert(words.end(), pk.data.begin(), pk.data.end());
  }
  abcHipCheck(abc_hip_load_relin_key(ctx, words.data()), "relinearisation key upload");
}

void HipCiphertextFactory::saveGaloisKeys(std::ostream &out, int compression) const {
  const size_t N = ciphertextSlotSize, K = chain.size(), per = 2 * K * N;
  sealwire::KSwitchImage img;
  img.id = sealwire::parmsId(wireParms(ckksMode, ciphertextSlotSize, chain, plainModulus), K);
  img.keys.resize(N);  // seal::GaloisKeys: entry (galois_elt - 1) / 2, empty where no key was generated
  std::vector<uint64_t> words((size_t)limbs * per);
  const int count = abc_hip_num_galois_keys(ctx);
  for (int k = 0; k < count; ++k) {
    const uint32_t elt = abc_hip_galois_elt_at(ctx, k);
    abcHipCheck(abc_hip_get_galois_key(ctx, elt, words.data()), "Galois key download");
    auto &entry = img.keys[(elt - 1) >> 1];
    for (int j = 0; j < limbs; ++j) entry.push_back(keyImage(img.id, N, K, words.data() + (size_t)j * per));
  }
  sealwire::save(out, img, wireMode(compression));
}
void HipCiphertextFactory::loadGaloisKeys(std::istream &in) {
  const size_t N = ciphertextSlotSize, K = chain.size(), per = 2 * K * N;
  sealwire::KSwitchImage img;
  sealwire::load(in, img, (uint64_t)64 * (limbs + 1) * per);
  const auto id = sealwire::parmsId(wireParms(ckksMode, ciphertextSlotSize, chain, plainModulus), K);
  if (img.id != id) throw std::runtime_error("loadGaloisKeys: parms_id is not this factory's key-level parameter set");
  if (img.keys.size() != N) throw std::runtime_error("loadGaloisKeys: expected N entries (one per odd Galois element below 2N)");
  std::vector<uint64_t> words;
  for (size_t idx = 0; idx < N; ++idx) {
    const auto &entry = img.keys[idx];
    if (entry.empty()) continue;
    if (entry.size() != (size_t)limbs) throw std::runtime_error("loadGaloisKeys: expected one key per data prime");
    words.clear();
    for (const auto &pk : entry) {
      checkKeyImage(pk, id, N, chain, "loadGaloisKeys");
      words.insert(words.end(), pk.data.begin(), pk.data.end());
    }
    abcHipCheck(abc_hip_load_galois_key(ctx, (uint32_t)(2 * idx + 1), words.data()), "Galois key upload");
  }
}
