#include "HipCiphertext.hpp"

#include <cmath>

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
// CKKS plain operand: integer or real cleartexts
std::vector<double> realCleartext(const ICleartext &operand, const char *op) {
  if (auto p = dynamic_cast<const Cleartext<int> *>(&operand)) return std::vector<double>(p->getData().begin(), p->getData().end());
  if (auto p = dynamic_cast<const Cleartext<double> *>(&operand)) return p->getData();
  if (auto p = dynamic_cast<const Cleartext<float> *>(&operand)) return std::vector<double>(p->getData().begin(), p->getData().end());
  throw std::runtime_error(std::string(op) + "(Ciphertext,Cleartext) requires a Cleartext<int>, <float> or <double> under CKKS.");
}
}  // namespace

HipCiphertext::Buffer::~Buffer() {
  if (p) abc_hip_free(f.context(), p);  // stream-ordered: every use of the buffer was issued before this point
}

std::shared_ptr<HipCiphertext::Buffer> HipCiphertext::allocate(const HipCiphertextFactory &f, int level) {
  void *p = nullptr;
  abcHipCheck(abc_hip_malloc(f.context(), &p, f.ciphertextWords(level) * 8), "ciphertext allocation");
  return std::make_shared<Buffer>(f, static_cast<uint64_t *>(p));
}

HipCiphertext::HipCiphertext(const std::reference_wrapper<const HipCiphertextFactory> hipFactory)
    : AbstractCiphertext((const std::reference_wrapper<const AbstractCiphertextFactory>)hipFactory),
      buf(allocate(hipFactory.get(), hipFactory.get().dataLimbs())), nl(hipFactory.get().dataLimbs()),
      sc(hipFactory.get().isCkks() ? hipFactory.get().defaultScale() : 1.0) {}

HipCiphertext::HipCiphertext(const std::reference_wrapper<const HipCiphertextFactory> hipFactory, int level, double scale)
    : AbstractCiphertext((const std::reference_wrapper<const AbstractCiphertextFactory>)hipFactory),
      buf(allocate(hipFactory.get(), level)), nl(level), sc(scale) {
  if (level < 1 || level > hipFactory.get().dataLimbs()) throw std::runtime_error("HipCiphertext: level out of range");
}

HipCiphertext::~HipCiphertext() = default;

// "deep copy" by value semantics: shares the buffer until one side writes
HipCiphertext::HipCiphertext(const HipCiphertext &other) : AbstractCiphertext(other.factory), buf(other.buf), nl(other.nl), sc(other.sc) {}

HipCiphertext::HipCiphertext(HipCiphertext &&other) noexcept
    : AbstractCiphertext(other.factory), buf(std::move(other.buf)), nl(other.nl), sc(other.sc) {}

HipCiphertext &HipCiphertext::operator=(const HipCiphertext &other) { return *this = HipCiphertext(other); }

HipCiphertext &HipCiphertext::operator=(HipCiphertext &&other) {
  if (&other == this) return *this;
  if (&factory.get() != &other.factory.get())
    throw std::runtime_error("Cannot move Ciphertext from factory A into Ciphertext created by Factory B.");
  buf = std::move(other.buf);
  nl = other.nl;
  sc = other.sc;
  return *this;
}

std::shared_ptr<HipCiphertext::Buffer> HipCiphertext::target() const {
  return buf.use_count() == 1 ? buf : allocate(getFactory(), nl);
}

uint64_t *HipCiphertext::devicePtr() {
  if (buf.use_count() != 1) {  // a writer gets its own copy
    auto t = allocate(getFactory(), nl);
    abcHipCheck(abc_hip_memcpy_d2d(getFactory().context(), t->p, buf->p, getFactory().ciphertextWords(nl) * 8), "clone");
    buf = std::move(t);
  }
  return buf->p;
}

// ---- CKKS level / scale management ----
void HipCiphertext::dropTo(int level) {
  const auto &f = getFactory();
  while (nl > level) {  // Evaluator::mod_switch_to_next: CKKS simply drops the last limb
    auto t = allocate(f, nl - 1);
    abcHipCheck(abc_hip_mod_switch(f.context(), in(), t->p, 2, nl, f.batchSize()), "mod_switch");
    buf = std::move(t);
    --nl;
  }
}
void HipCiphertext::rescaleIfPossible() {
  const auto &f = getFactory();
  if (nl < 2) return;  // nothing left to divide by: the scale stays squared (documented in INTEGRATION.md)
  auto t = allocate(f, nl - 1);
  abcHipCheck(abc_hip_rescale(f.context(), in(), t->p, 2, nl, f.batchSize()), "rescale");
  sc /= (double)f.prime(nl - 1);
  buf = std::move(t);
  --nl;
}
void HipCiphertext::checkScales(double a, double b) {
  // after a rescale the scale is Delta^2 / q_l, a hair off Delta for a 40-bit prime next to Delta = 2^40 (relative 1e-7 .. 1e-6
  // for the first NTT primes below a power of two): adding such values is what every CKKS program does (SEAL makes the user
  // overwrite the scale; here the left operand's scale stands).  Anything coarser is a real mismatch.  Written so that a NaN or
  // a non-positive scale fails too.
  const double rel = std::fabs(a - b) / std::fmax(std::fabs(a), std::fabs(b));
  if (!(a > 0.0) || !(b > 0.0) || !(rel <= 1e-5))
    throw std::runtime_error("CKKS: scale mismatch between operands (" + std::to_string(a) + " vs " + std::to_string(b) + ")");
}
// a product's scale must leave room under the modulus of its level, or the message wraps around silently (SEAL throws
// "scale out of bounds" at the same point)
void HipCiphertext::checkScaleFits(double scale, int level) const {
  const auto &f = getFactory();
  double bits = 0.0;
  for (int j = 0; j < level; ++j) bits += std::log2((double)f.prime(j));
  if (!std::isfinite(scale) || !(scale > 0.0) || !(std::log2(scale) < bits - 1.0))
    throw std::runtime_error("CKKS: scale 2^" + std::to_string(std::log2(scale)) + " out of bounds for a modulus of " + std::to_string(bits) +
                             " bits (no limb left to rescale into)");
}
const uint64_t *HipCiphertext::alignWith(const HipCiphertext &operand, std::shared_ptr<Buffer> &keep) {
  const auto &f = getFactory();
  if (!f.isCkks() || operand.nl == nl) return operand.in();
  if (operand.nl < nl) {  // we are the one to come down
    dropTo(operand.nl);
    return operand.in();
  }
  // the operand is const: bring a temporary copy of it down to our level
  const uint64_t *src = operand.in();
  for (int l = operand.nl; l > nl; --l) {
    auto t = allocate(f, l - 1);
    abcHipCheck(abc_hip_mod_switch(f.context(), src, t->p, 2, l, f.batchSize()), "mod_switch");
    keep = std::move(t);
    src = keep->p;
  }
  return src;
}

const HipCiphertextFactory &HipCiphertext::getFactory() const {
  if (auto f = dynamic_cast<const HipCiphertextFactory *>(&factory.get())) return *f;
  throw std::runtime_error("Cast of AbstractFactory to HipFactory failed. HipCiphertext is probably invalid.");
}

std::unique_ptr<HipCiphertext> HipCiphertext::fresh(int level) const {
  auto r = std::make_unique<HipCiphertext>(*this);  // same factory, level and scale bookkeeping
  r->buf = allocate(getFactory(), level);
  r->nl = level;
  return r;
}
std::unique_ptr<HipCiphertext> HipCiphertext::clone_impl() const { return std::make_unique<HipCiphertext>(*this); }
std::unique_ptr<AbstractCiphertext> HipCiphertext::clone() const { return clone_impl(); }

int HipCiphertext::noiseBits() const {
  throw std::runtime_error("noiseBits: invariant noise budget is a host-side diagnostic not provided by the HIP backend.");
}

// ---- ctxt-ctxt ----
// returning forms = copy-on-write clone + in-place form (a clone is a reference; the in-place form on a shared buffer
// computes out of place into a fresh one, so no device-to-device copy is made either way)
std::unique_ptr<AbstractCiphertext> HipCiphertext::add(const AbstractCiphertext &operand) const {
  auto r = clone_impl();
  r->addInplace(operand);
  return r;
}
std::unique_ptr<AbstractCiphertext> HipCiphertext::subtract(const AbstractCiphertext &operand) const {
  auto r = clone_impl();
  r->subtractInplace(operand);
  return r;
}
std::unique_ptr<AbstractCiphertext> HipCiphertext::multiply(const AbstractCiphertext &operand) const {
  auto r = clone_impl();
  r->multiplyInplace(operand);
  return r;
}
void HipCiphertext::addInplace(const AbstractCiphertext &operand) {
  std::shared_ptr<Buffer> keep;
  const uint64_t *rhs = alignWith(cast(operand), keep);
  if (getFactory().isCkks()) checkScales(sc, cast(operand).sc);
  auto t = target();
  abcHipCheck(abc_hip_add(getFactory().context(), in(), rhs, t->p, 2, nl, getFactory().batchSize()), "add");
  adopt(std::move(t));
}
void HipCiphertext::subtractInplace(const AbstractCiphertext &operand) {
  std::shared_ptr<Buffer> keep;
  const uint64_t *rhs = alignWith(cast(operand), keep);
  if (getFactory().isCkks()) checkScales(sc, cast(operand).sc);
  auto t = target();
  abcHipCheck(abc_hip_sub(getFactory().context(), in(), rhs, t->p, 2, nl, getFactory().batchSize()), "sub");
  adopt(std::move(t));
}
void HipCiphertext::multiplyInplace(const AbstractCiphertext &operand) {
  // Evaluator::multiply + relinearize_inplace, src/runtime/SealCiphertext.cpp:102-107 (CKKS: + rescale_to_next)
  std::shared_ptr<Buffer> keep;
  const uint64_t *rhs = alignWith(cast(operand), keep);
  auto t = target();
  abcHipCheck(abc_hip_mul_relin(getFactory().context(), in(), rhs, t->p, nl, getFactory().batchSize()), "multiply");
  adopt(std::move(t));
  if (getFactory().isCkks()) {
    sc *= cast(operand).sc;
    checkScaleFits(sc, nl);
    rescaleIfPossible();
  }
}

// ---- rotation ----
std::unique_ptr<AbstractCiphertext> HipCiphertext::rotateRows(int steps) const {
  auto r = fresh(nl);
  abcHipCheck(abc_hip_rotate(getFactory().context(), in(), r->buf->p, nl, steps, getFactory().batchSize()), "rotate_rows");
  return r;
}
void HipCiphertext::rotateRowsInplace(int steps) {
  auto t = target();
  abcHipCheck(abc_hip_rotate(getFactory().context(), in(), t->p, nl, steps, getFactory().batchSize()), "rotate_rows");
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
  if (getFactory().isCkks()) {  // encoded at this value's own level and scale: exact scale match
    const uint64_t *pl = getFactory().cachedCkksPlaintext(realCleartext(operand, "ADD"), nl, sc);
    auto t = target();
    abcHipCheck(abc_hip_add_plain(getFactory().context(), in(), pl, 0, t->p, 2, nl, getFactory().batchSize()), "add_plain");
    adopt(std::move(t));
    return;
  }
  DevicePlain pl(getFactory(), intCleartext(operand, "ADD").getData());
  auto t = target();
  abcHipCheck(abc_hip_add_plain(getFactory().context(), in(), pl.p, 0, t->p, 2, nl, getFactory().batchSize()), "add_plain");
  adopt(std::move(t));
}
void HipCiphertext::subtractPlainInplace(const ICleartext &operand) {
  if (getFactory().isCkks()) {
    const uint64_t *pl = getFactory().cachedCkksPlaintext(realCleartext(operand, "SUB"), nl, sc);
    auto t = target();
    abcHipCheck(abc_hip_sub_plain(getFactory().context(), in(), pl, 0, t->p, 2, nl, getFactory().batchSize()), "sub_plain");
    adopt(std::move(t));
    return;
  }
  DevicePlain pl(getFactory(), intCleartext(operand, "SUB").getData());
  auto t = target();
  abcHipCheck(abc_hip_sub_plain(getFactory().context(), in(), pl.p, 0, t->p, 2, nl, getFactory().batchSize()), "sub_plain");
  adopt(std::move(t));
}
void HipCiphertext::multiplyPlainInplace(const ICleartext &operand) {
  if (getFactory().isCkks()) {
    const auto vals = realCleartext(operand, "MULTIPLY");
    bool allMinusOne = !vals.empty();
    for (double v : vals) allMinusOne = allMinusOne && v == -1.0;
    auto t = target();
    if (allMinusOne) {  // negation shortcut, src/runtime/SealCiphertext.cpp:192-193: no level is spent
      abcHipCheck(abc_hip_negate(getFactory().context(), in(), t->p, 2, nl, getFactory().batchSize()), "negate");
      adopt(std::move(t));
      return;
    }
    const double ps = getFactory().defaultScale();
    const uint64_t *pl = getFactory().cachedCkksPlaintext(vals, nl, ps);
    abcHipCheck(abc_hip_multiply_plain(getFactory().context(), in(), pl, 0, t->p, 2, nl, getFactory().batchSize()), "multiply_plain");
    adopt(std::move(t));
    sc *= ps;
    checkScaleFits(sc, nl);
    rescaleIfPossible();
    return;
  }
  const auto &ct = intCleartext(operand, "MULTIPLY");
  if (ct.allEqual(-1)) {  // negation shortcut, src/runtime/SealCiphertext.cpp:192-193
    auto t = target();
    abcHipCheck(abc_hip_negate(getFactory().context(), in(), t->p, 2, nl, getFactory().batchSize()), "negate");
    adopt(std::move(t));
    return;
  }
  DevicePlain pl(getFactory(), ct.getData());
  // multiply_plain keeps size 2, so the reference's relinearize_inplace (:197) is a no-op
  auto t = target();
  abcHipCheck(abc_hip_multiply_plain(getFactory().context(), in(), pl.p, 0, t->p, 2, nl, getFactory().batchSize()),
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
