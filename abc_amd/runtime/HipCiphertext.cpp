#include "HipCiphertext.hpp"

#include "../../include/abc_hip.h"
#include "HipCiphertextFactory.hpp"

void abcHipCheck(int status, const char *what) {
  if (status != 0) throw std::runtime_error(std::string(what) + ": " + abc_hip_last_error());
}

namespace {
const HipCiphertext &cast(const AbstractCiphertext &c) {
  if (auto p = dynamic_cast<const HipCiphertext *>(&c)) return *p;
  throw std::runtime_error("Cast of AbstractCiphertext to HipCiphertext failed!");
}
const Cleartext<int> &intCleartext(const ICleartext &operand, const char *op) {
  if (auto p = dynamic_cast<const Cleartext<int> *>(&operand)) return *p;
  throw std::runtime_error(std::string(op) + "(Ciphertext,Cleartext) requires a Cleartext<int> as BFV supports integers only.");
}
struct DevicePlain {  // encoded plain operand, from the factory's cache (the reference re-encodes it on every plain op)
  const uint64_t *p;
  DevicePlain(const HipCiphertextFactory &fac, const std::vector<int> &v) : p(fac.cachedPlaintext(v)) {}
};
}  // namespace

HipCiphertext::Buffer::~Buffer() {
  if (p) abc_hip_free(f.context(), p);  // stream-ordered: every use of the buffer was issued before this point
}

std::shared_ptr<HipCiphertext::Buffer> HipCiphertext::allocate(const HipCiphertextFactory &f) {
  void *p = nullptr;
  abcHipCheck(abc_hip_malloc(f.context(), &p, f.ciphertextWords() * 8), "ciphertext allocation");
  return std::make_shared<Buffer>(f, static_cast<uint64_t *>(p));
}

HipCiphertext::HipCiphertext(const std::reference_wrapper<const HipCiphertextFactory> hipFactory)
    : AbstractCiphertext((const std::reference_wrapper<const AbstractCiphertextFactory>)hipFactory), buf(allocate(hipFactory.get())) {}

HipCiphertext::~HipCiphertext() = default;

// "deep copy" by value semantics: shares the buffer until one side writes
HipCiphertext::HipCiphertext(const HipCiphertext &other) : AbstractCiphertext(other.factory), buf(other.buf) {}

HipCiphertext::HipCiphertext(HipCiphertext &&other) noexcept : AbstractCiphertext(other.factory), buf(std::move(other.buf)) {}

HipCiphertext &HipCiphertext::operator=(const HipCiphertext &other) { return *this = HipCiphertext(other); }

HipCiphertext &HipCiphertext::operator=(HipCiphertext &&other) {
  if (&other == this) return *this;
  if (&factory.get() != &other.factory.get())
    throw std::runtime_error("Cannot move Ciphertext from factory A into Ciphertext created by Factory B.");
  buf = std::move(other.buf);
  return *this;
}

std::shared_ptr<HipCiphertext::Buffer> HipCiphertext::target() const {
  return buf.use_count() == 1 ? buf : allocate(getFactory());
}

uint64_t *HipCiphertext::devicePtr() {
  if (buf.use_count() != 1) {  // a writer gets its own copy
    auto t = allocate(getFactory());
    abcHipCheck(abc_hip_memcpy_d2d(getFactory().context(), t->p, buf->p, getFactory().ciphertextWords() * 8), "clone");
    buf = std::move(t);
  }
  return buf->p;
}

const HipCiphertextFactory &HipCiphertext::getFactory() const {
  if (auto f = dynamic_cast<const HipCiphertextFactory *>(&factory.get())) return *f;
  throw std::runtime_error("Cast of AbstractFactory to HipFactory failed. HipCiphertext is probably invalid.");
}

std::unique_ptr<HipCiphertext> HipCiphertext::fresh() const { return std::make_unique<HipCiphertext>(std::cref(getFactory())); }
std::unique_ptr<HipCiphertext> HipCiphertext::clone_impl() const { return std::make_unique<HipCiphertext>(*this); }
std::unique_ptr<AbstractCiphertext> HipCiphertext::clone() const { return clone_impl(); }

int HipCiphertext::noiseBits() const {
  throw std::runtime_error("noiseBits: invariant noise budget is a host-side diagnostic not provided by the HIP backend.");
}

// ---- ctxt-ctxt ----
std::unique_ptr<AbstractCiphertext> HipCiphertext::add(const AbstractCiphertext &operand) const {
  auto r = fresh();
  abcHipCheck(abc_hip_add(getFactory().context(), in(), cast(operand).in(), r->buf->p, 2, getFactory().dataLimbs(), getFactory().batchSize()), "add");
  return r;
}
std::unique_ptr<AbstractCiphertext> HipCiphertext::subtract(const AbstractCiphertext &operand) const {
  auto r = fresh();
  abcHipCheck(abc_hip_sub(getFactory().context(), in(), cast(operand).in(), r->buf->p, 2, getFactory().dataLimbs(), getFactory().batchSize()), "sub");
  return r;
}
std::unique_ptr<AbstractCiphertext> HipCiphertext::multiply(const AbstractCiphertext &operand) const {
  // Evaluator::multiply + relinearize_inplace, src/runtime/SealCiphertext.cpp:102-107
  auto r = fresh();
  abcHipCheck(abc_hip_mul_relin(getFactory().context(), in(), cast(operand).in(), r->buf->p, getFactory().dataLimbs(), getFactory().batchSize()),
              "multiply");
  return r;
}
void HipCiphertext::addInplace(const AbstractCiphertext &operand) {
  auto t = target();
  abcHipCheck(abc_hip_add(getFactory().context(), in(), cast(operand).in(), t->p, 2, getFactory().dataLimbs(), getFactory().batchSize()), "add");
  adopt(std::move(t));
}
void HipCiphertext::subtractInplace(const AbstractCiphertext &operand) {
  auto t = target();
  abcHipCheck(abc_hip_sub(getFactory().context(), in(), cast(operand).in(), t->p, 2, getFactory().dataLimbs(), getFactory().batchSize()), "sub");
  adopt(std::move(t));
}
void HipCiphertext::multiplyInplace(const AbstractCiphertext &operand) {
  auto t = target();
  abcHipCheck(abc_hip_mul_relin(getFactory().context(), in(), cast(operand).in(), t->p, getFactory().dataLimbs(), getFactory().batchSize()),
              "multiply");
  adopt(std::move(t));
}

// ---- rotation ----
std::unique_ptr<AbstractCiphertext> HipCiphertext::rotateRows(int steps) const {
  auto r = fresh();
  abcHipCheck(abc_hip_rotate(getFactory().context(), in(), r->buf->p, getFactory().dataLimbs(), steps, getFactory().batchSize()), "rotate_rows");
  return r;
}
void HipCiphertext::rotateRowsInplace(int steps) {
  auto t = target();
  abcHipCheck(abc_hip_rotate(getFactory().context(), in(), t->p, getFactory().dataLimbs(), steps, getFactory().batchSize()), "rotate_rows");
  adopt(std::move(t));
}

// ---- ctxt-plain ----
std::unique_ptr<AbstractCiphertext> HipCiphertext::addPlain(const ICleartext &operand) const {
  auto r = clone_impl();
  r->addPlainInplace(operand);
  return r;
}
std::unique_ptr<AbstractCiphertext> HipCiphertext::subtractPlain(const ICleartext &operand) const {
  auto r = clone_impl();
  r->subtractPlainInplace(operand);
  return r;
}
std::unique_ptr<AbstractCiphertext> HipCiphertext::multiplyPlain(const ICleartext &operand) const {
  auto r = clone_impl();
  r->multiplyPlainInplace(operand);
  return r;
}
void HipCiphertext::addPlainInplace(const ICleartext &operand) {
  DevicePlain pl(getFactory(), intCleartext(operand, "ADD").getData());
  auto t = target();
  abcHipCheck(abc_hip_add_plain(getFactory().context(), in(), pl.p, 0, t->p, 2, getFactory().dataLimbs(), getFactory().batchSize()), "add_plain");
  adopt(std::move(t));
}
void HipCiphertext::subtractPlainInplace(const ICleartext &operand) {
  DevicePlain pl(getFactory(), intCleartext(operand, "SUB").getData());
  auto t = target();
  abcHipCheck(abc_hip_sub_plain(getFactory().context(), in(), pl.p, 0, t->p, 2, getFactory().dataLimbs(), getFactory().batchSize()), "sub_plain");
  adopt(std::move(t));
}
void HipCiphertext::multiplyPlainInplace(const ICleartext &operand) {
  const auto &ct = intCleartext(operand, "MULTIPLY");
  if (ct.allEqual(-1)) {  // negation shortcut, src/runtime/SealCiphertext.cpp:192-193
    auto t = target();
    abcHipCheck(abc_hip_negate(getFactory().context(), in(), t->p, 2, getFactory().dataLimbs(), getFactory().batchSize()), "negate");
    adopt(std::move(t));
    return;
  }
  DevicePlain pl(getFactory(), ct.getData());
  // multiply_plain keeps size 2, so the reference's relinearize_inplace (:197) is a no-op
  auto t = target();
  abcHipCheck(abc_hip_multiply_plain(getFactory().context(), in(), pl.p, 0, t->p, 2, getFactory().dataLimbs(), getFactory().batchSize()),
              "multiply_plain");
  adopt(std::move(t));
}

// ---- AbstractValue dispatch (src/runtime/SealCiphertext.cpp:208-239) ----
void HipCiphertext::add_inplace(const AbstractValue &other) {
  if (auto c = dynamic_cast<const HipCiphertext *>(&other)) addInplace(*c);
  else if (auto p = dynamic_cast<const ICleartext *>(&other)) addPlainInplace(*p);
  else throw std::runtime_error("Operation ADD only supported for (AbstractCiphertext,AbstractCiphertext) and (HipCiphertext, ICleartext).");
}
void HipCiphertext::subtract_inplace(const AbstractValue &other) {
  if (auto c = dynamic_cast<const HipCiphertext *>(&other)) subtractInplace(*c);
  else if (auto p = dynamic_cast<const ICleartext *>(&other)) subtractPlainInplace(*p);
  else throw std::runtime_error("Operation SUBTRACT only supported for (HipCiphertext,HipCiphertext) and (HipCiphertext, ICleartext).");
}
void HipCiphertext::multiply_inplace(const AbstractValue &other) {
  if (auto c = dynamic_cast<const HipCiphertext *>(&other)) multiplyInplace(*c);
  else if (auto p = dynamic_cast<const ICleartext *>(&other)) multiplyPlainInplace(*p);
  else throw std::runtime_error("Operation MULTIPLY only supported for (HipCiphertext,HipCiphertext) and (HipCiphertext, ICleartext).");
}
#define ABC_UNSUPPORTED(name) \
  void HipCiphertext::name(const AbstractValue &) { throw std::runtime_error("Operation " #name " not supported for (HipCiphertext, ANY)."); }
ABC_UNSUPPORTED(divide_inplace)
ABC_UNSUPPORTED(modulo_inplace)
ABC_UNSUPPORTED(logicalAnd_inplace)
ABC_UNSUPPORTED(logicalOr_inplace)
ABC_UNSUPPORTED(logicalLess_inplace)
ABC_UNSUPPORTED(logicalLessEqual_inplace)
ABC_UNSUPPORTED(logicalGreater_inplace)
ABC_UNSUPPORTED(logicalGreaterEqual_inplace)
ABC_UNSUPPORTED(logicalEqual_inplace)
ABC_UNSUPPORTED(logicalNotEqual_inplace)
ABC_UNSUPPORTED(bitwiseAnd_inplace)
ABC_UNSUPPORTED(bitwiseXor_inplace)
ABC_UNSUPPORTED(bitwiseOr_inplace)
#undef ABC_UNSUPPORTED
void HipCiphertext::logicalNot_inplace() {
  throw std::runtime_error("Operation logicalNot_inplace not supported for (HipCiphertext, ANY). For an arithmetic negation, multiply_inplace by (-1) instead.");
}
void HipCiphertext::bitwiseNot_inplace() {
  throw std::runtime_error("Operation bitwiseNot_inplace not supported for (HipCiphertext, ANY). For an arithmetic negation, multiply_inplace by (-1) instead.");
}
