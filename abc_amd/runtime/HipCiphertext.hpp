// HipCiphertext -- an AbstractCiphertext whose data lives in MI355X HBM and whose every operation is a call
// into libabc_hip.so (include/abc_hip.h).  Drop-in counterpart of the reference's SealCiphertext
// (include/ast_opt/runtime/SealCiphertext.h:14-112, src/runtime/SealCiphertext.cpp): same methods, same
// operator set (add / subtract / multiply supported, the other 15 AbstractValue operators throw
// std::runtime_error, src/runtime/SealCiphertext.cpp:241-309), same ownership (unique_ptr to the
// ciphertext, non-owning reference to the factory, cross-factory move-assign throws, :25-34).
#pragma once

#include <memory>

#include "plugin_api.hpp"

#include "HipCiphertextFactory.hpp"

class HipCiphertext : public AbstractCiphertext {
  // Device buffer [B][2][L][N] (B = the factory's batch size, 1 by default), shared copy-on-write: clone() and the
  // copy constructor only take another reference (the interpreter clones on EVERY variable read,
  // src/runtime/RuntimeVisitor.cpp:431-437, and most clones are only ever read), and an in-place operation on a shared
  // buffer computes out of place into a fresh one -- every C-ABI operation takes separate input and output pointers --
  // so value semantics hold without a single device-to-device copy.
  struct Buffer {
    const HipCiphertextFactory &f;
    uint64_t *p;
    Buffer(const HipCiphertextFactory &fac, uint64_t *ptr) : f(fac), p(ptr) {}
    ~Buffer();
    Buffer(const Buffer &) = delete;
    Buffer &operator=(const Buffer &) = delete;
  };
  std::shared_ptr<Buffer> buf;
  // CKKS bookkeeping (BFV: nl = L, sc = 1): data limbs the value currently has and its scale.  seal::Ciphertext carries
  // the same two (parms_id, scale); here the plugin classes also act on them, so the interpreter needs no CKKS knowledge:
  // products are rescaled while a limb can be dropped, operands at different levels meet at the lower one.
  int nl = 0;
  double sc = 1.0;

  static std::shared_ptr<Buffer> allocate(const HipCiphertextFactory &f, int level);
  std::unique_ptr<HipCiphertext> clone_impl() const;
  std::unique_ptr<HipCiphertext> fresh(int level) const;
  uint64_t *in() const { return buf->p; }
  // destination of an in-place operation: the own buffer if nobody shares it, otherwise a fresh one, which `adopt`
  // installs after the operation has been issued
  std::shared_ptr<Buffer> target() const;
  void adopt(std::shared_ptr<Buffer> t) { buf = std::move(t); }
  // CKKS helpers
  void dropTo(int level);                       // mod_switch down (no-op at or below `level`)
  void rescaleIfPossible();                     // divide by the last prime of the current level
  static void checkScales(double a, double b);  // additions need (nearly) equal scales
  void checkScaleFits(double scale, int level) const;  // a product's scale must stay below the modulus of its level
  // this and operand at a common level: returns the operand's pointer (possibly a temporary held in `keep`)
  const uint64_t *alignWith(const HipCiphertext &operand, std::shared_ptr<Buffer> &keep);

 public:
  ~HipCiphertext() override;
  explicit HipCiphertext(const std::reference_wrapper<const HipCiphertextFactory> hipFactory);
  // uninitialised value at a given level and scale (the factory fills it: loadCiphertext)
  HipCiphertext(const std::reference_wrapper<const HipCiphertextFactory> hipFactory, int level, double scale);
  HipCiphertext(const HipCiphertext &other);
  HipCiphertext(HipCiphertext &&other) noexcept;
  HipCiphertext &operator=(const HipCiphertext &other);
  HipCiphertext &operator=(HipCiphertext &&other);

  [[nodiscard]] const uint64_t *devicePtr() const { return buf->p; }
  [[nodiscard]] int level() const { return nl; }
  [[nodiscard]] double scale() const { return sc; }
  [[nodiscard]] uint64_t *devicePtr();  // for writing: un-shares first
  [[nodiscard]] const HipCiphertextFactory &getFactory() const override;
  [[nodiscard]] int noiseBits() const;  // SealCiphertext::noiseBits, SealCiphertext.cpp:80-83 (host-side diagnostic)

  std::unique_ptr<AbstractCiphertext> multiply(const AbstractCiphertext &operand) const override;
  void multiplyInplace(const AbstractCiphertext &operand) override;
  std::unique_ptr<AbstractCiphertext> multiplyPlain(const ICleartext &operand) const override;
  void multiplyPlainInplace(const ICleartext &operand) override;
  std::unique_ptr<AbstractCiphertext> add(const AbstractCiphertext &operand) const override;
  void addInplace(const AbstractCiphertext &operand) override;
  std::unique_ptr<AbstractCiphertext> addPlain(const ICleartext &operand) const override;
  void addPlainInplace(const ICleartext &operand) override;
  std::unique_ptr<AbstractCiphertext> subtract(const AbstractCiphertext &operand) const override;
  void subtractInplace(const AbstractCiphertext &operand) override;
  std::unique_ptr<AbstractCiphertext> subtractPlain(const ICleartext &operand) const override;
  void subtractPlainInplace(const ICleartext &operand) override;
  std::unique_ptr<AbstractCiphertext> rotateRows(int steps) const override;
  void rotateRowsInplace(int steps) override;
  std::unique_ptr<AbstractCiphertext> clone() const override;

  void add_inplace(const AbstractValue &other) override;
  void subtract_inplace(const AbstractValue &other) override;
  void multiply_inplace(const AbstractValue &other) override;
  void divide_inplace(const AbstractValue &other) override;
  void modulo_inplace(const AbstractValue &other) override;
  void logicalAnd_inplace(const AbstractValue &other) override;
  void logicalOr_inplace(const AbstractValue &other) override;
  void logicalLess_inplace(const AbstractValue &other) override;
  void logicalLessEqual_inplace(const AbstractValue &other) override;
  void logicalGreater_inplace(const AbstractValue &other) override;
  void logicalGreaterEqual_inplace(const AbstractValue &other) override;
  void logicalEqual_inplace(const AbstractValue &other) override;
  void logicalNotEqual_inplace(const AbstractValue &other) override;
  void logicalNot_inplace() override;
  void bitwiseAnd_inplace(const AbstractValue &other) override;
  void bitwiseXor_inplace(const AbstractValue &other) override;
  void bitwiseOr_inplace(const AbstractValue &other) override;
  void bitwiseNot_inplace() override;
};
