// plugin_api.hpp -- the value / ciphertext / factory plugin surface of ABC's runtime, restated so that
// this repository builds and tests stand-alone.
//
// Inside the ABC tree define ABC_HIP_USE_REFERENCE_HEADERS: the shim then includes ABC's own headers
//   include/ast_opt/runtime/AbstractValue.h:4-48, AbstractCiphertext.h:12-99,
//   include/ast_opt/runtime/AbstractCiphertextFactory.h:13-50, Cleartext.h:13-27,30-223
// and this file contributes nothing.  Names, signatures, const-ness and error behaviour
// (std::runtime_error for everything, src/runtime/SealCiphertext.cpp:241-309) are those of the reference,
// so HipCiphertext / HipCiphertextFactory are source-compatible drop-ins for SealCiphertext /
// SealCiphertextFactory.
#pragma once

#ifdef ABC_HIP_USE_REFERENCE_HEADERS
#include "ast_opt/runtime/AbstractCiphertext.h"
#include "ast_opt/runtime/AbstractCiphertextFactory.h"
#include "ast_opt/runtime/AbstractValue.h"
#include "ast_opt/runtime/Cleartext.h"
#else

#include <algorithm>
#include <cstdint>
#include <functional>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

class AbstractCiphertext;
class AbstractValue;

// Everything the interpreter manipulates: 18 in-place operators (only +,-,* are supported on ciphertexts).
class AbstractValue {
 protected:
  AbstractValue() = default;

 public:
  virtual ~AbstractValue() = default;
#define ABC_BINARY_OP(name) virtual void name(const AbstractValue &other) = 0;
  ABC_BINARY_OP(add_inplace)
  ABC_BINARY_OP(subtract_inplace)
  ABC_BINARY_OP(multiply_inplace)
  ABC_BINARY_OP(divide_inplace)
  ABC_BINARY_OP(modulo_inplace)
  ABC_BINARY_OP(logicalAnd_inplace)
  ABC_BINARY_OP(logicalOr_inplace)
  ABC_BINARY_OP(logicalLess_inplace)
  ABC_BINARY_OP(logicalLessEqual_inplace)
  ABC_BINARY_OP(logicalGreater_inplace)
  ABC_BINARY_OP(logicalGreaterEqual_inplace)
  ABC_BINARY_OP(logicalEqual_inplace)
  ABC_BINARY_OP(logicalNotEqual_inplace)
  ABC_BINARY_OP(bitwiseAnd_inplace)
  ABC_BINARY_OP(bitwiseXor_inplace)
  ABC_BINARY_OP(bitwiseOr_inplace)
#undef ABC_BINARY_OP
  virtual void logicalNot_inplace() = 0;
  virtual void bitwiseNot_inplace() = 0;
};

class AbstractCiphertextFactory {
 public:
  virtual ~AbstractCiphertextFactory() = default;
  virtual std::unique_ptr<AbstractCiphertext> createCiphertext(const std::vector<int64_t> &data) const = 0;
  virtual std::unique_ptr<AbstractCiphertext> createCiphertext(const std::vector<int> &data) const = 0;
  virtual std::unique_ptr<AbstractCiphertext> createCiphertext(int64_t data) const = 0;
  virtual std::unique_ptr<AbstractCiphertext> createCiphertext(std::unique_ptr<AbstractValue> &&cleartext) const = 0;
  virtual void decryptCiphertext(AbstractCiphertext &abstractCiphertext, std::vector<int64_t> &ciphertextData) const = 0;
  virtual std::string getString(AbstractCiphertext &abstractCiphertext) const = 0;
};

class ICleartext : public AbstractValue {
 protected:
  ICleartext() = default;

 public:
  ~ICleartext() override = default;
  virtual std::unique_ptr<ICleartext> clone() = 0;
  virtual std::string toString() = 0;
  virtual void setValueAtIndex(int idx, std::unique_ptr<AbstractValue> &&newValue) = 0;
};

class AbstractCiphertext : public AbstractValue {
 protected:
  explicit AbstractCiphertext(const std::reference_wrapper<const AbstractCiphertextFactory> acf) : factory(acf) {}
  const std::reference_wrapper<const AbstractCiphertextFactory> factory;  // non-owning: factory outlives ciphertexts

 public:
  ~AbstractCiphertext() override = default;
  [[nodiscard]] virtual std::unique_ptr<AbstractCiphertext> multiply(const AbstractCiphertext &operand) const = 0;
  virtual void multiplyInplace(const AbstractCiphertext &operand) = 0;
  [[nodiscard]] virtual std::unique_ptr<AbstractCiphertext> multiplyPlain(const ICleartext &operand) const = 0;
  virtual void multiplyPlainInplace(const ICleartext &operand) = 0;
  [[nodiscard]] virtual std::unique_ptr<AbstractCiphertext> add(const AbstractCiphertext &operand) const = 0;
  virtual void addInplace(const AbstractCiphertext &operand) = 0;
  [[nodiscard]] virtual std::unique_ptr<AbstractCiphertext> addPlain(const ICleartext &operand) const = 0;
  virtual void addPlainInplace(const ICleartext &operand) = 0;
  [[nodiscard]] virtual std::unique_ptr<AbstractCiphertext> subtract(const AbstractCiphertext &operand) const = 0;
  virtual void subtractInplace(const AbstractCiphertext &operand) = 0;
  [[nodiscard]] virtual std::unique_ptr<AbstractCiphertext> subtractPlain(const ICleartext &operand) const = 0;
  virtual void subtractPlainInplace(const ICleartext &operand) = 0;
  [[nodiscard]] virtual std::unique_ptr<AbstractCiphertext> rotateRows(int steps) const = 0;
  virtual void rotateRowsInplace(int steps) = 0;
  virtual std::unique_ptr<AbstractCiphertext> clone() const = 0;
  [[nodiscard]] virtual const AbstractCiphertextFactory &getFactory() const { return factory; }
};

// Cleartext<T>: a vector of public values with element-wise operators.  (The reference can also build it
// from AST Literal / ExpressionList nodes; those constructors belong to the AST layer and are not
// needed behind the plugin boundary.)
template <typename T>
class Cleartext : public ICleartext {
  std::vector<T> data;

  template <class F>
  void zipWith(F f, const AbstractValue &other) {
    auto rhs = dynamic_cast<const Cleartext<T> *>(&other);
    if (!rhs)
      throw std::runtime_error(
          "Given operation can only be applied on (Cleartext<T>, Cleartext<T>). This could happen, for example, if an "
          "operation is called on (Cleartext<T>, AbstractCiphertext) but the operation is unsupported in FHE.");
    std::transform(data.begin(), data.end(), rhs->data.begin(), data.begin(), f);
  }

 public:
  explicit Cleartext(const std::vector<T> values) : data(values) {}
  explicit Cleartext(std::vector<std::unique_ptr<ICleartext>> &parts) {
    for (auto &p : parts) {
      auto same = dynamic_cast<Cleartext<T> *>(p.get());
      if (!same) throw std::runtime_error("Cannot create Cleartext<T> of multiple other Cleartext<T> with different types!");
      data.insert(data.end(), same->data.begin(), same->data.end());
    }
  }
  // Cleartext<bool> from a Cleartext<int> (result of a relational operator)
  explicit Cleartext(std::unique_ptr<AbstractValue> &&value) {
    auto ints = dynamic_cast<Cleartext<int> *>(value.get());
    if (!std::is_same<T, bool>::value || !ints)
      throw std::runtime_error("This constructor is only defined to take Cleartext<int> and generate a Cleartext<bool>.");
    for (int v : ints->getData()) data.push_back(static_cast<T>(v));
  }
  Cleartext(const Cleartext<T> &other) : ICleartext(), data(other.data) {}

  [[nodiscard]] bool allEqual(T value) const {
    return std::all_of(data.begin(), data.end(), [&](const T &v) { return v == value; });
  }
  [[nodiscard]] bool allEqual() const { return allEqual(data.at(0)); }
  [[nodiscard]] const std::vector<T> &getData() const { return data; }

  std::unique_ptr<ICleartext> clone() override { return std::make_unique<Cleartext<T>>(*this); }
  std::string toString() override {
    std::ostringstream os;
    for (size_t i = 0; i + 1 < data.size(); ++i) os << data[i] << ", ";
    os << data.back();
    return os.str();
  }
  void setValueAtIndex(int idx, std::unique_ptr<AbstractValue> &&newValue) override {
    auto v = dynamic_cast<Cleartext<T> *>(newValue.get());
    if (!v) throw std::runtime_error("Assigning a value to a Cleartext<T> requires the value to be a Cleartext<T> too (i.e., same type T).");
    if (!v->allEqual()) throw std::runtime_error("Cannot assign multiple values to a single Cleartext element.");
    data[idx] = v->getData().at(0);
  }

  // public - secret, done right: encrypt the public operand with the ciphertext's factory, subtract, RETURN the ciphertext.
  // Upstream's Cleartext<int>::subtract_inplace builds exactly this value and then drops it, because an in-place operator
  // on a Cleartext cannot turn the receiver into a ciphertext (include/ast_opt/runtime/Cleartext.h:349-360): its `1 --- c`
  // silently leaves the cleartext unchanged.  Deviation, documented in INTEGRATION.md: here the in-place form refuses
  // (it cannot deliver the result) and this method delivers it; CircuitRuntime uses it for `public - secret`.
  [[nodiscard]] std::unique_ptr<AbstractCiphertext> subtractCiphertext(const AbstractCiphertext &secret) const {
    if constexpr (std::is_same<T, int>::value) {
      std::unique_ptr<AbstractCiphertext> lhs = secret.getFactory().createCiphertext(data);
      lhs->subtractInplace(secret);
      return lhs;
    } else {
      throw std::runtime_error("public - secret is defined for Cleartext<int> only.");
    }
  }

  void add_inplace(const AbstractValue &o) override { zipWith(std::plus<T>(), o); }
  void subtract_inplace(const AbstractValue &o) override {
    if (dynamic_cast<const AbstractCiphertext *>(&o))
      throw std::runtime_error(
          "Cleartext - AbstractCiphertext cannot be computed in place (the result is a ciphertext): use "
          "Cleartext<int>::subtractCiphertext, which returns it.");
    zipWith(std::minus<T>(), o);
  }
  void multiply_inplace(const AbstractValue &o) override { zipWith(std::multiplies<T>(), o); }
  void divide_inplace(const AbstractValue &o) override {
    if constexpr (std::is_same<T, bool>::value) throw std::invalid_argument("Cannot divide_inplace booleans.");
    else zipWith(std::divides<T>(), o);
  }
  void modulo_inplace(const AbstractValue &o) override {
    if constexpr (std::is_same<T, bool>::value) throw std::invalid_argument("Cannot modulo_inplace booleans.");
    else if constexpr (std::is_floating_point<T>::value) throw std::runtime_error("Cannot apply modulo to operands of a floating-point type.");
    else zipWith(std::modulus<T>(), o);
  }
  void logicalAnd_inplace(const AbstractValue &o) override { zipWith(std::logical_and<T>(), o); }
  void logicalOr_inplace(const AbstractValue &o) override { zipWith(std::logical_or<T>(), o); }
  void logicalLess_inplace(const AbstractValue &o) override { zipWith(std::less<T>(), o); }
  void logicalLessEqual_inplace(const AbstractValue &o) override { zipWith(std::less_equal<T>(), o); }
  void logicalGreater_inplace(const AbstractValue &o) override { zipWith(std::greater<T>(), o); }
  void logicalGreaterEqual_inplace(const AbstractValue &o) override { zipWith(std::greater_equal<T>(), o); }
  void logicalEqual_inplace(const AbstractValue &o) override { zipWith(std::equal_to<T>(), o); }
  void logicalNotEqual_inplace(const AbstractValue &o) override { zipWith(std::not_equal_to<T>(), o); }
  void bitwiseAnd_inplace(const AbstractValue &o) override {
    if constexpr (std::is_floating_point<T>::value) throw std::runtime_error("Cannot apply a bitwise operator to operands of a floating-point type.");
    else zipWith(std::bit_and<T>(), o);
  }
  void bitwiseXor_inplace(const AbstractValue &o) override {
    if constexpr (std::is_floating_point<T>::value) throw std::runtime_error("Cannot apply a bitwise operator to operands of a floating-point type.");
    else zipWith(std::bit_xor<T>(), o);
  }
  void bitwiseOr_inplace(const AbstractValue &o) override {
    if constexpr (std::is_floating_point<T>::value) throw std::runtime_error("Cannot apply a bitwise operator to operands of a floating-point type.");
    else zipWith(std::bit_or<T>(), o);
  }
  void logicalNot_inplace() override { for (auto it = data.begin(); it != data.end(); ++it) *it = !*it; }
  void bitwiseNot_inplace() override {
    if constexpr (std::is_same<T, bool>::value) { for (auto it = data.begin(); it != data.end(); ++it) *it = !*it; }
    else if constexpr (std::is_floating_point<T>::value) throw std::runtime_error("Cannot apply bitwise-NOT to an operand of a floating-point type.");
    else { for (auto &v : data) v = ~v; }
  }
};

#endif  // ABC_HIP_USE_REFERENCE_HEADERS
