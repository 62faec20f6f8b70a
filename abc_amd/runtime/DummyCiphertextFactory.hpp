// DummyCiphertextFactory / DummyCiphertext -- the cleartext stand-in backend "for faster testing"
// (reference README.md:75-76; include/ast_opt/runtime/DummyCiphertext.h, DummyCiphertextFactory.h,
// src/runtime/DummyCiphertext.cpp, src/runtime/DummyCiphertextFactory.cpp).  BASELINE.json config 1
// ("element-wise add of two length-4 secret vectors, dummy (cleartext) runtime on CPU -- plumbing, no GPU")
// runs on it through the same CircuitRuntime as the HIP backend.  Reference semantics kept: values are
// int64, no padding, operand sizes must match ("Sizes of data vectors do not match",
// DummyCiphertext.cpp:49-63), rotateRows throws "Not yet implemented." (:244-249).
#pragma once

#include "plugin_api.hpp"

class DummyCiphertextFactory : public AbstractCiphertextFactory {
 public:
  std::unique_ptr<AbstractCiphertext> createCiphertext(const std::vector<int64_t> &data) const override;
  std::unique_ptr<AbstractCiphertext> createCiphertext(const std::vector<int> &data) const override;
  std::unique_ptr<AbstractCiphertext> createCiphertext(int64_t data) const override;
  std::unique_ptr<AbstractCiphertext> createCiphertext(std::unique_ptr<AbstractValue> &&cleartext) const override;
  void decryptCiphertext(AbstractCiphertext &abstractCiphertext, std::vector<int64_t> &ciphertextData) const override;
  std::string getString(AbstractCiphertext &abstractCiphertext) const override;
};

class DummyCiphertext : public AbstractCiphertext {
  std::vector<int64_t> values;

  template <class F>
  void zip(const std::vector<int64_t> &rhs, F f) {
    if (rhs.size() != values.size()) throw std::runtime_error("Sizes of data vectors do not match");
    for (size_t i = 0; i < values.size(); ++i) values[i] = f(values[i], rhs[i]);
  }
  static const std::vector<int64_t> &dataOf(const AbstractCiphertext &c);
  static std::vector<int64_t> dataOf(const ICleartext &c, const char *op);
  std::unique_ptr<DummyCiphertext> copy() const { return std::make_unique<DummyCiphertext>(*this); }
  [[noreturn]] static void unsupported(const char *name);

 public:
  explicit DummyCiphertext(const std::reference_wrapper<const AbstractCiphertextFactory> f) : AbstractCiphertext(f) {}
  DummyCiphertext(const DummyCiphertext &o) : AbstractCiphertext(o.factory), values(o.values) {}
  void createFresh(const std::vector<int64_t> &data) { values = data; }
  const std::vector<int64_t> &getData() const { return values; }

  std::unique_ptr<AbstractCiphertext> multiply(const AbstractCiphertext &o) const override { auto r = copy(); r->multiplyInplace(o); return r; }
  void multiplyInplace(const AbstractCiphertext &o) override { zip(dataOf(o), [](int64_t a, int64_t b) { return a * b; }); }
  std::unique_ptr<AbstractCiphertext> multiplyPlain(const ICleartext &o) const override { auto r = copy(); r->multiplyPlainInplace(o); return r; }
  void multiplyPlainInplace(const ICleartext &o) override { zip(dataOf(o, "Multiply"), [](int64_t a, int64_t b) { return a * b; }); }
  std::unique_ptr<AbstractCiphertext> add(const AbstractCiphertext &o) const override { auto r = copy(); r->addInplace(o); return r; }
  void addInplace(const AbstractCiphertext &o) override { zip(dataOf(o), [](int64_t a, int64_t b) { return a + b; }); }
  std::unique_ptr<AbstractCiphertext> addPlain(const ICleartext &o) const override { auto r = copy(); r->addPlainInplace(o); return r; }
  void addPlainInplace(const ICleartext &o) override { zip(dataOf(o, "ADD"), [](int64_t a, int64_t b) { return a + b; }); }
  std::unique_ptr<AbstractCiphertext> subtract(const AbstractCiphertext &o) const override { auto r = copy(); r->subtractInplace(o); return r; }
  void subtractInplace(const AbstractCiphertext &o) override { zip(dataOf(o), [](int64_t a, int64_t b) { return a - b; }); }
  std::unique_ptr<AbstractCiphertext> subtractPlain(const ICleartext &o) const override { auto r = copy(); r->subtractPlainInplace(o); return r; }
  void subtractPlainInplace(const ICleartext &o) override { zip(dataOf(o, "SUB"), [](int64_t a, int64_t b) { return a - b; }); }
  std::unique_ptr<AbstractCiphertext> rotateRows(int) const override { throw std::runtime_error("Not yet implemented."); }
  void rotateRowsInplace(int) override { throw std::runtime_error("Not yet implemented."); }
  std::unique_ptr<AbstractCiphertext> clone() const override { return copy(); }

  void add_inplace(const AbstractValue &other) override;
  void subtract_inplace(const AbstractValue &other) override;
  void multiply_inplace(const AbstractValue &other) override;
  void divide_inplace(const AbstractValue &) override { unsupported("divide_inplace"); }
  void modulo_inplace(const AbstractValue &) override { unsupported("modulo_inplace"); }
  void logicalAnd_inplace(const AbstractValue &) override { unsupported("logicalAnd_inplace"); }
  void logicalOr_inplace(const AbstractValue &) override { unsupported("logicalOr_inplace"); }
  void logicalLess_inplace(const AbstractValue &) override { unsupported("logicalLess_inplace"); }
  void logicalLessEqual_inplace(const AbstractValue &) override { unsupported("logicalLessEqual_inplace"); }
  void logicalGreater_inplace(const AbstractValue &) override { unsupported("logicalGreater_inplace"); }
  void logicalGreaterEqual_inplace(const AbstractValue &) override { unsupported("logicalGreaterEqual_inplace"); }
  void logicalEqual_inplace(const AbstractValue &) override { unsupported("logicalEqual_inplace"); }
  void logicalNotEqual_inplace(const AbstractValue &) override { unsupported("logicalNotEqual_inplace"); }
  void logicalNot_inplace() override { unsupported("logicalNot_inplace"); }
  void bitwiseAnd_inplace(const AbstractValue &) override { unsupported("bitwiseAnd_inplace"); }
  void bitwiseXor_inplace(const AbstractValue &) override { unsupported("bitwiseXor_inplace"); }
  void bitwiseOr_inplace(const AbstractValue &) override { unsupported("bitwiseOr_inplace"); }
  void bitwiseNot_inplace() override { unsupported("bitwiseNot_inplace"); }
};
