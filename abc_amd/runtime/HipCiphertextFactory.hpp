// HipCiphertextFactory -- AbstractCiphertextFactory over libabc_hip.so.  Drop-in counterpart of
// SealCiphertextFactory (include/ast_opt/runtime/SealCiphertextFactory.h:13-120,
// src/runtime/SealCiphertextFactory.cpp): BFV, N = numElementsPerCiphertextSlot (default 16384, :16),
// coefficient modulus BFVDefault(N) (:80), plaintext modulus Batching(N, 20) (:83), secret / public /
// relinearisation / all default Galois keys (:89-93), pad-with-last-value (:102-115).
#pragma once

#include <deque>
#include <iosfwd>
#include <string>
#include <vector>

#include "CkksEncoder.hpp"
#include "GraphCapable.hpp"
#include "plugin_api.hpp"

struct abc_hip_ctx;

// Scheme policy of a HipCiphertextFactory.  The default is the reference's own choice (BFV, BFVDefault(N), Batching(N, 20):
// SealCiphertextFactory.cpp:72-100).  CKKS is the hook the reference left open (CMakeLists.txt:216, HAVE_SEAL_CKKS): a
// modulus chain given by bit sizes (data limbs, then the special prime) and a default scale.  The caller of the plugin
// surface needs nothing else: levels and scales are tracked per ciphertext, ct x ct and ct x plain products are rescaled
// (one limb dropped) while limbs remain, and operands at different levels are brought to the lower one.
struct HipSchemeConfig {
  bool ckks = false;
  unsigned int ringDegree = 16'384;
  std::vector<int> ckksBits = {50, 40, 40, 40, 50};  // the benchmark chain
  double ckksScale = 1099511627776.0;                // 2^40
  int device = 0;
  uint64_t seed = 0;  // 0: keys and encryption randomness from the OS-keyed generator; else the reproducible TEST spec
  size_t batch = 1;
};

class HipCiphertextFactory : public AbstractCiphertextFactory, public GraphCapable {
  const unsigned int ciphertextSlotSize = 16'384;
  abc_hip_ctx *ctx = nullptr;  // owned: device tables + keys
  int limbs = 0;               // data limbs L
  uint64_t keySeed = 0;
  // Batch mode (an extension the reference lacks; it evaluates one circuit at a time, RuntimeVisitor.h:34): every
  // ciphertext of this factory is B independent ciphertexts that all operations process in one batched device call, so
  // ONE pass of an unchanged interpreter (ABC's RuntimeVisitor or CircuitRuntime) evaluates the circuit on B input sets.
  size_t batch = 1;
  mutable std::deque<std::vector<std::vector<int64_t>>> queuedInputs;
  mutable std::deque<std::vector<std::vector<double>>> queuedRealInputs;
  // Encoded plaintexts of recent plain operands (the reference re-encodes its operand on every plain operation,
  // SealCiphertext.cpp:132,143,154: here a repeated constant costs one encode + upload, and no device synchronisation)
  struct CachedPlain {
    std::vector<int64_t> values;
    uint64_t *d_plain;
  };
  mutable std::deque<CachedPlain> plainCache;
  static constexpr size_t kPlainCacheEntries = 32;

  // CKKS policy (empty / unused for BFV)
  bool ckksMode = false;
  double ckksScale = 0;
  std::vector<int> ckksBits;
  std::vector<uint64_t> chain;  // the context's primes: data limbs, then the special prime
  uint64_t plainModulus = 0;    // BFV t (0 for CKKS)
  CkksEncoder ckksEncoder;
  struct CachedCkksPlain {
    std::vector<double> values;
    int level;
    double scale;
    uint64_t *d_plain;
  };
  mutable std::deque<CachedCkksPlain> ckksPlainCache;

  void setupContext(int device);
  template <typename T>
  std::vector<T> expandVector(const std::vector<T> &values) const;
  std::unique_ptr<AbstractCiphertext> createCkksCiphertext(const std::vector<double> &data) const;
  int encryptInto(const void *d_plain, uint64_t *d_ct) const;

 public:
  HipCiphertextFactory();
  explicit HipCiphertextFactory(unsigned int numElementsPerCiphertextSlot, int device = 0, uint64_t seed = 0, size_t batchSize = 1);
  explicit HipCiphertextFactory(const HipSchemeConfig &config);
  virtual ~HipCiphertextFactory();  // (ABC's AbstractCiphertextFactory declares no virtual destructor)
  HipCiphertextFactory(const HipCiphertextFactory &) = delete;  // one device context per factory
  HipCiphertextFactory &operator=(const HipCiphertextFactory &) = delete;

  [[nodiscard]] abc_hip_ctx *context() const { return ctx; }
  [[nodiscard]] unsigned int getCiphertextSlotSize() const { return ciphertextSlotSize; }
  [[nodiscard]] int dataLimbs() const { return limbs; }
  [[nodiscard]] size_t batchSize() const { return batch; }
  [[nodiscard]] size_t ciphertextWords() const { return batch * 2 * limbs * ciphertextSlotSize; }  // of one (batched) value
  [[nodiscard]] size_t ciphertextWords(int level) const { return batch * 2 * (size_t)level * ciphertextSlotSize; }
  [[nodiscard]] bool isCkks() const { return ckksMode; }
  [[nodiscard]] double defaultScale() const { return ckksScale; }
  [[nodiscard]] uint64_t prime(int j) const { return chain[j]; }
  // slots a value may fill: N for BFV (two rows of N/2), N/2 for CKKS (one row; rotateRows rotates it cyclically)
  [[nodiscard]] unsigned int usableSlots() const { return ckksMode ? ciphertextSlotSize / 2 : ciphertextSlotSize; }
  // CKKS: device plaintext [level][N] (NTT form) of public values at a given level and scale, from the factory's cache
  const uint64_t *cachedCkksPlaintext(const std::vector<double> &value, int level, double scale) const;
  // CKKS: all slot values of instance 0 as doubles (decryptCiphertext rounds them to the interface's int64)
  void decryptCiphertextReal(AbstractCiphertext &abstractCiphertext, std::vector<double> &out) const;
  void decryptCiphertextRealBatch(AbstractCiphertext &abstractCiphertext, std::vector<std::vector<double>> &out) const;
  // batch mode for real-valued inputs (CKKS)
  void queueBatchedRealInput(std::vector<std::vector<double>> perInstance) const;
  std::unique_ptr<AbstractCiphertext> createCiphertext(const std::vector<double> &data) const;

  // Batch mode: the next createCiphertext call encrypts these B vectors (one per circuit instance) instead of B copies
  // of its argument; calls are served in queue order, i.e. in the order the interpreter declares its secret inputs.
  void queueBatchedInput(std::vector<std::vector<int64_t>> perInstance) const;
  // all B decrypted slot vectors of a batched ciphertext (decryptCiphertext returns instance 0, the reference's view)
  void decryptCiphertextBatch(AbstractCiphertext &abstractCiphertext, std::vector<std::vector<int64_t>> &out) const;

  // device plaintext [N] (coefficients mod t) from public values; caller frees with freeDevice
  uint64_t *createPlaintext(const std::vector<int> &value) const;
  uint64_t *createPlaintext(const std::vector<int64_t> &value) const;
  uint64_t *createPlaintext(int64_t value) const;
  void freeDevice(void *p) const;
  // same, owned by the factory's cache: valid until kPlainCacheEntries further distinct operands have been encoded
  // (release is stream-ordered, so operations already issued on an evicted plaintext stay correct)
  const uint64_t *cachedPlaintext(const std::vector<int> &value) const;

  std::unique_ptr<AbstractCiphertext> createCiphertext(const std::vector<int64_t> &data) const override;
  std::unique_ptr<AbstractCiphertext> createCiphertext(const std::vector<int> &data) const override;
  std::unique_ptr<AbstractCiphertext> createCiphertext(int64_t data) const override;
  std::unique_ptr<AbstractCiphertext> createCiphertext(std::unique_ptr<AbstractValue> &&abstractValue) const override;
  void decryptCiphertext(AbstractCiphertext &abstractCiphertext, std::vector<int64_t> &ciphertextData) const override;
  std::string getString(AbstractCiphertext &abstractCiphertext) const override;

  // ---- SEAL 3.6 wire format (SealWire.hpp: layout, provenance; PARITY UNPINNED -- no SEAL in this image) ----
  // The reference serialises nothing; this is the additive row f3 of SURVEY.md section 8: a client may keep seal::KeyGenerator /
  // Encryptor / Decryptor and hand evaluation to this backend.  compression: 0 none, 1 zlib, 2 zstd (seal::compr_mode_type).
  // A batched value is written / read as its B instances, one SEAL object after the other.
  void saveCiphertext(const AbstractCiphertext &ciphertext, std::ostream &out, int compression = 0) const;
  std::unique_ptr<AbstractCiphertext> loadCiphertext(std::istream &in) const;
  void saveSecretKey(std::ostream &out, int compression = 0) const;
  void savePublicKey(std::ostream &out, int compression = 0) const;
  void saveRelinKeys(std::ostream &out, int compression = 0) const;
  void saveGaloisKeys(std::ostream &out, int compression = 0) const;
  // replace this factory's keys by SEAL-generated ones (same parameters, checked through parms_id); seed-compressed objects
  // (Serializable<...>) are refused
  void loadSecretKey(std::istream &in);
  void loadPublicKey(std::istream &in);
  void loadRelinKeys(std::istream &in);
  void loadGaloisKeys(std::istream &in);

  // GraphCapable: recorded circuits (abc_hip_graph_*)
  void graphBegin() const override;
  void *graphEnd() const override;
  void graphAbort() const override;
  void graphLaunch(void *graph) const override;
  void graphDestroy(void *graph) const override;
  void synchronize() const override;
  void rewriteCiphertext(AbstractCiphertext &target, const std::vector<int64_t> &values) const override;
  void rewriteCiphertextBatch(AbstractCiphertext &target, const std::vector<std::vector<int64_t>> &perInstance) const override;
};

// maps a non-zero C-ABI status to the reference's error convention (std::runtime_error)
void abcHipCheck(int status, const char *what);
