#include "DummyCiphertextFactory.hpp"

const std::vector<int64_t> &DummyCiphertext::dataOf(const AbstractCiphertext &c) {
  if (auto d = dynamic_cast<const DummyCiphertext *>(&c)) return d->values;
  throw std::runtime_error("Cast of AbstractCiphertext to DummyCiphertext failed!");
}
std::vector<int64_t> DummyCiphertext::dataOf(const ICleartext &c, const char *op) {
  if (auto ints = dynamic_cast<const Cleartext<int> *>(&c)) return std::vector<int64_t>(ints->getData().begin(), ints->getData().end());
  throw std::runtime_error(std::string(op) + "(Ciphertext,Cleartext) requires a Cleartext<int> as BFV supports integers only.");
}
void DummyCiphertext::unsupported(const char *name) {
  throw std::runtime_error(std::string("Operation ") + name + " not supported for (DummyCiphertext, ANY).");
}
void DummyCiphertext::add_inplace(const AbstractValue &other) {
  if (auto c = dynamic_cast<const DummyCiphertext *>(&other)) addInplace(*c);
  else if (auto p = dynamic_cast<const ICleartext *>(&other)) addPlainInplace(*p);
  else throw std::runtime_error("Operation ADD only supported for (AbstractCiphertext,AbstractCiphertext) and (DummyCiphertext, ICleartext).");
}
void DummyCiphertext::subtract_inplace(const AbstractValue &other) {
  if (auto c = dynamic_cast<const DummyCiphertext *>(&other)) subtractInplace(*c);
  else if (auto p = dynamic_cast<const ICleartext *>(&other)) subtractPlainInplace(*p);
  else throw std::runtime_error("Operation SUBTRACT only supported for (DummyCiphertext,DummyCiphertext) and (DummyCiphertext, ICleartext).");
}
void DummyCiphertext::multiply_inplace(const AbstractValue &other) {
  if (auto c = dynamic_cast<const DummyCiphertext *>(&other)) multiplyInplace(*c);
  else if (auto p = dynamic_cast<const ICleartext *>(&other)) multiplyPlainInplace(*p);
  else throw std::runtime_error("Operation MULTIPLY only supported for (DummyCiphertext,DummyCiphertext) and (DummyCiphertext, ICleartext).");
}

std::unique_ptr<AbstractCiphertext> DummyCiphertextFactory::createCiphertext(const std::vector<int64_t> &data) const {
  auto c = std::make_unique<DummyCiphertext>(std::cref(static_cast<const AbstractCiphertextFactory &>(*this)));
  c->createFresh(data);
  return c;
}
std::unique_ptr<AbstractCiphertext> DummyCiphertextFactory::createCiphertext(const std::vector<int> &data) const {
  return createCiphertext(std::vector<int64_t>(data.begin(), data.end()));
}
std::unique_ptr<AbstractCiphertext> DummyCiphertextFactory::createCiphertext(int64_t data) const {
  return createCiphertext(std::vector<int64_t>{data});
}
std::unique_ptr<AbstractCiphertext> DummyCiphertextFactory::createCiphertext(std::unique_ptr<AbstractValue> &&cleartext) const {
  if (auto ints = dynamic_cast<Cleartext<int> *>(cleartext.get()))
    return createCiphertext(std::vector<int64_t>(ints->getData().begin(), ints->getData().end()));
  throw std::runtime_error(
      "Cannot create ciphertext from any other than a Cleartext<int> as used ciphertext factory (DummyCiphertextFactory) uses BFV "
      "that only supports integers.");
}
void DummyCiphertextFactory::decryptCiphertext(AbstractCiphertext &abstractCiphertext, std::vector<int64_t> &ciphertextData) const {
  auto d = dynamic_cast<DummyCiphertext *>(&abstractCiphertext);
  if (!d) throw std::runtime_error("Cast of AbstractCiphertext to DummyCiphertext failed!");
  ciphertextData = d->getData();
}
std::string DummyCiphertextFactory::getString(AbstractCiphertext &abstractCiphertext) const {
  std::vector<int64_t> values;
  decryptCiphertext(abstractCiphertext, values);
  std::stringstream ss;
  ss << "[";
  for (const auto v : values) ss << " " << v << ", ";
  ss.seekp(-1, ss.cur);
  ss << " ]";
  return ss.str();
}
