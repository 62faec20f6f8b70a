#include "CircuitRuntime.hpp"

#include <cctype>
#include <iostream>

#include "GraphCapable.hpp"

namespace {
typedef CircuitRuntime::Token Token;

std::vector<Token> tokenize(const std::string &s) {
  std::vector<Token> out;
  size_t i = 0;
  static const char *three[] = {"+++", "---", "***"};
  static const char *two[] = {"<=", ">=", "==", "!="};
  while (i < s.size()) {
    const char ch = s[i];
    if (std::isspace((unsigned char)ch)) { ++i; continue; }
    if (ch == '/' && i + 1 < s.size() && s[i + 1] == '/') {  // line comment
      while (i < s.size() && s[i] != '\n') ++i;
      continue;
    }
    Token t;
    if (std::isalpha((unsigned char)ch) || ch == '_') {
      size_t j = i;
      while (j < s.size() && (std::isalnum((unsigned char)s[j]) || s[j] == '_')) ++j;
      t.kind = Token::Ident;
      t.text = s.substr(i, j - i);
      i = j;
    } else if (std::isdigit((unsigned char)ch)) {
      size_t j = i;
      while (j < s.size() && std::isdigit((unsigned char)s[j])) ++j;
      t.kind = Token::Int;
      t.text = s.substr(i, j - i);
      t.value = std::stol(t.text);
      i = j;
    } else {
      t.kind = Token::Punct;
      bool matched = false;
      for (const char *p : three)
        if (s.compare(i, 3, p) == 0) { t.text = p; i += 3; matched = true; break; }
      if (!matched)
        for (const char *p : two)
          if (s.compare(i, 2, p) == 0) { t.text = p; i += 2; matched = true; break; }
      if (!matched) { t.text = std::string(1, ch); ++i; }
    }
    out.push_back(t);
  }
  out.push_back(Token{});
  return out;
}

bool isCiphertext(const AbstractValue *v) { return dynamic_cast<const AbstractCiphertext *>(v) != nullptr; }
}  // namespace

void CircuitRuntime::load(const std::string &src) {
  toks = tokenize(src);
  pos = 0;
}
const Token &CircuitRuntime::peek(size_t ahead) const { return toks[std::min(pos + ahead, toks.size() - 1)]; }
Token CircuitRuntime::next() {
  Token t = peek();
  if (pos + 1 < toks.size()) ++pos;
  return t;
}
bool CircuitRuntime::accept(const std::string &p) {
  if (peek().kind == Token::Punct && peek().text == p) { next(); return true; }
  return false;
}
void CircuitRuntime::expect(const std::string &p) {
  if (!accept(p)) throw std::runtime_error("Parse error: expected '" + p + "' but found '" + peek().text + "'.");
}

CircuitRuntime::CircuitRuntime(AbstractCiphertextFactory &f, const std::string &inputs) : factory(f) {
  // the input block may only hold variable declarations (checkAstStructure<VariableDeclaration>, RuntimeVisitor.cpp:446-471)
  load(inputs);
  while (peek().kind != Token::End) statement(true);
}

void CircuitRuntime::executeAst(const std::string &program) {
  load(program);
  try {
    while (peek().kind != Token::End) statement(false);
  } catch (ReturnReached &) {
    std::cout << "Program reached return statement.." << std::endl;  // RuntimeVisitor.cpp:485
  }
}

void CircuitRuntime::statement(bool inputsOnly) {
  const Token t = peek();
  if (t.kind == Token::Ident && (t.text == "secret" || t.text == "int")) {
    bool secret = false;
    if (t.text == "secret") { next(); secret = true; }
    if (!(peek().kind == Token::Ident && peek().text == "int"))
      throw std::runtime_error("Only (secret) int vectors are supported by this runtime.");
    next();
    declaration(secret);
    expect(";");
    return;
  }
  if (inputsOnly) throw std::runtime_error("Block statements of given (in-/out)put AST must be of type VariableDeclaration. ");
  if (t.kind == Token::Ident && t.text == "for") { forLoop(); return; }
  if (t.kind == Token::Ident && t.text == "return") {
    next();
    if (!accept(";")) { expression(); expect(";"); }
    throw ReturnReached();
  }
  if (t.kind == Token::Ident && (t.text == "if" || t.text == "while" || t.text == "public"))
    throw std::runtime_error("Statement '" + t.text + "' is not supported by this runtime.");
  if (t.kind == Token::Punct && t.text == "{") { block(); return; }
  assignment();
  expect(";");
}

void CircuitRuntime::block() {
  expect("{");
  while (!(peek().kind == Token::Punct && peek().text == "}")) {
    if (peek().kind == Token::End) throw std::runtime_error("Parse error: unterminated block.");
    statement(false);
  }
  expect("}");
}

void CircuitRuntime::store(const std::string &name, bool secretVar, std::unique_ptr<AbstractValue> value, bool declare) {
  Variable &v = vars[name];
  if (declare) v.secret = secretVar;
  if (v.secret) {
    if (isCiphertext(value.get())) {
      v.ctxt.reset(dynamic_cast<AbstractCiphertext *>(value.release()));
    } else if (declare) {
      v.ctxt = factory.createCiphertext(std::move(value));  // encrypt the public initialiser
    } else {
      throw std::runtime_error("castUniquePtr failed: Cannot cast given unique_ptr from type AbstractValue to type AbstractCiphertext.");
    }
    v.clear.reset();
  } else {
    auto clear = dynamic_cast<ICleartext *>(value.get());
    if (!clear)
      throw std::runtime_error("Initialization value of VariableDeclaration ( " + name + ") could not be processed successfully.");
    value.release();
    v.clear.reset(clear);
    v.ctxt.reset();
  }
}

void CircuitRuntime::declaration(bool secret) {
  const Token id = next();
  if (id.kind != Token::Ident) throw std::runtime_error("Parse error: identifier expected in declaration.");
  if (!accept("="))
    throw std::runtime_error("Unsupported: Variable declaration without initializer encountered. Please specify an initialization value!");
  store(id.text, secret, expression(), true);
}

void CircuitRuntime::assignment() {
  const Token id = next();
  if (id.kind != Token::Ident) throw std::runtime_error("Assignments currently only supported to (non-indexed) variables.");
  if (!vars.count(id.text)) throw std::runtime_error("Assignment to undeclared variable '" + id.text + "'.");
  if (accept("[")) {  // i[2] = value on a public vector (RuntimeVisitor.cpp:362-381)
    auto idx = expression();
    expect("]");
    expect("=");
    auto value = expression();
    auto idxInt = dynamic_cast<Cleartext<int> *>(idx.get());
    if (!idxInt) throw std::runtime_error("Index given in IndexAccess must be an integer!");
    if (!idxInt->allEqual()) throw std::runtime_error("Index of IndexAccess must be a scalar.");
    Variable &v = vars[id.text];
    if (v.secret) throw std::runtime_error("Only simple, non-nested IndexAccesses on non-secret variables are supported yet (e.g., i[2] -> ok, i[j[2]] -> not supported).");
    v.clear->setValueAtIndex(idxInt->getData().at(0), std::move(value));
    return;
  }
  expect("=");
  auto value = expression();
  Variable &v = vars[id.text];
  // assigning a ciphertext to a variable makes it a ciphertext variable (RuntimeVisitor.cpp:345-349)
  if (isCiphertext(value.get())) v.secret = true;
  store(id.text, v.secret, std::move(value), false);
}

void CircuitRuntime::skipStatementOrBlock() {
  int depth = 0;
  for (;;) {
    const Token t = next();
    if (t.kind == Token::End) throw std::runtime_error("Parse error: unexpected end of program.");
    if (t.kind != Token::Punct) continue;
    if (t.text == "{" || t.text == "(") ++depth;
    else if (t.text == "}" || t.text == ")") { if (--depth == 0 && t.text == "}") return; }
  }
}

void CircuitRuntime::forLoop() {
  next();  // for
  expect("(");
  if (!accept(";")) {  // initializer
    if (peek().kind == Token::Ident && (peek().text == "int" || peek().text == "secret")) {
      bool secret = false;
      if (peek().text == "secret") { next(); secret = true; }
      next();
      declaration(secret);
    } else {
      assignment();
    }
    expect(";");
  }
  const size_t condPos = pos;
  if (peek().kind == Token::Punct && peek().text == ";") throw std::runtime_error("For loops without a condition are not supported yet!");
  int depth = 0;  // skip the condition
  while (!(depth == 0 && peek().kind == Token::Punct && peek().text == ";")) {
    if (peek().kind == Token::End) throw std::runtime_error("Parse error in for header.");
    if (peek().text == "(") ++depth;
    if (peek().text == ")") --depth;
    next();
  }
  expect(";");
  const size_t updPos = pos;
  depth = 0;  // skip the update
  while (!(depth == 0 && peek().kind == Token::Punct && peek().text == ")")) {
    if (peek().kind == Token::End) throw std::runtime_error("Parse error in for header.");
    if (peek().text == "(") ++depth;
    if (peek().text == ")") --depth;
    next();
  }
  const bool hasUpdate = (pos != updPos);
  expect(")");
  const size_t bodyPos = pos;
  skipStatementOrBlock();
  const size_t endPos = pos;
  for (;;) {
    pos = condPos;
    auto cond = expression();
    if (isCiphertext(cond.get())) throw std::runtime_error("For loops over secret conditions are not supported yet!");
    auto asBool = dynamic_cast<Cleartext<bool> *>(cond.get());
    if (!asBool) throw std::runtime_error("For loop's condition must be evaluable to a Boolean.");
    if (!(asBool->allEqual() && asBool->getData().front())) break;
    pos = bodyPos;
    block();
    if (hasUpdate) { pos = updPos; assignment(); }
  }
  pos = endPos;
}

std::unique_ptr<AbstractValue> CircuitRuntime::readVariable(const std::string &name) {
  auto it = vars.find(name);
  if (it == vars.end()) throw std::runtime_error("Identifier '" + name + "' cannot be resolved.");
  // the maps own the value and it may be referenced again later: clone
  if (it->second.secret) return it->second.ctxt->clone();
  return it->second.clear->clone();
}

std::unique_ptr<AbstractValue> CircuitRuntime::expression() { return relational(); }

std::unique_ptr<AbstractValue> CircuitRuntime::relational() {
  auto lhs = additive();
  while (peek().kind == Token::Punct &&
         (peek().text == "<" || peek().text == "<=" || peek().text == ">" || peek().text == ">=" || peek().text == "==" ||
          peek().text == "!=")) {
    const std::string op = next().text;
    lhs = binary(op, std::move(lhs), additive());
  }
  return lhs;
}
std::unique_ptr<AbstractValue> CircuitRuntime::additive() {
  auto lhs = multiplicative();
  while (peek().kind == Token::Punct && (peek().text == "+" || peek().text == "-" || peek().text == "+++" || peek().text == "---")) {
    const std::string op = next().text;
    lhs = binary(op, std::move(lhs), multiplicative());
  }
  return lhs;
}
std::unique_ptr<AbstractValue> CircuitRuntime::multiplicative() {
  auto lhs = primary();
  while (peek().kind == Token::Punct && (peek().text == "*" || peek().text == "***" || peek().text == "/" || peek().text == "%")) {
    const std::string op = next().text;
    lhs = binary(op, std::move(lhs), primary());
  }
  return lhs;
}

std::unique_ptr<AbstractValue> CircuitRuntime::primary() {
  const Token t = next();
  if (t.kind == Token::Int) return std::make_unique<Cleartext<int>>(std::vector<int>{(int)t.value});
  if (t.kind == Token::Punct && t.text == "-") {
    const Token n = next();
    if (n.kind != Token::Int) throw std::runtime_error("Parse error: integer literal expected after '-'.");
    return std::make_unique<Cleartext<int>>(std::vector<int>{-(int)n.value});
  }
  if (t.kind == Token::Punct && t.text == "(") {
    auto v = expression();
    expect(")");
    return v;
  }
  if (t.kind == Token::Punct && t.text == "{") {  // expression list of literals
    std::vector<int> values;
    if (!accept("}")) {
      do {
        auto e = expression();
        auto lit = dynamic_cast<Cleartext<int> *>(e.get());
        if (!lit) throw std::runtime_error("Found ExpressionList that does contain any other than ICleartext element. Aborting...");
        values.insert(values.end(), lit->getData().begin(), lit->getData().end());
      } while (accept(","));
      expect("}");
    }
    return std::make_unique<Cleartext<int>>(values);
  }
  if (t.kind == Token::Ident && t.text == "rotate") {
    expect("(");
    // arg 0 must be a plain variable, arg 1 an integer literal (RuntimeVisitor.cpp:139-154)
    const Token id = next();
    if (id.kind != Token::Ident || !(peek().kind == Token::Punct && peek().text == ","))
      throw std::runtime_error("Argument 'ciphertext' in 'rotate' instruction must be a variable.");
    expect(",");
    bool neg = accept("-");
    const Token st = next();
    if (st.kind != Token::Int || !(peek().kind == Token::Punct && peek().text == ")"))
      throw std::runtime_error("Argument 'steps' in 'rotate' instruction must be an integer.");
    expect(")");
    auto it = vars.find(id.text);
    if (it == vars.end() || !it->second.secret) throw std::runtime_error("rotate: '" + id.text + "' is not a declared ciphertext.");
    return it->second.ctxt->rotateRows(neg ? -(int)st.value : (int)st.value);  // out of place, no clone
  }
  if (t.kind == Token::Ident) {
    if (peek().kind == Token::Punct && peek().text == "(")
      throw std::runtime_error("Calls other than 'rotate(identifier: label, numSteps: int);' are not supported yet!");
    if (accept("[")) {  // index access on a public vector (RuntimeVisitor.cpp:268-298)
      auto it = vars.find(t.text);
      if (it != vars.end() && it->second.secret)
        throw std::runtime_error("IndexAccess for secret variables is not supported by RuntimeVisitor. This should have already been removed by the Vectorizer. Error?");
      auto target = readVariable(t.text);
      auto idx = expression();
      expect("]");
      auto tv = dynamic_cast<Cleartext<int> *>(target.get());
      auto iv = dynamic_cast<Cleartext<int> *>(idx.get());
      if (!tv || !iv) throw std::runtime_error("IndexAccess only implemented for Cleartext<int> yet.");
      if (!iv->allEqual()) throw std::runtime_error("The resolved index of the IndexAccess doesn't seem like to be a scalar integer.");
      return std::make_unique<Cleartext<int>>(std::vector<int>{tv->getData().at(iv->getData().at(0))});
    }
    return readVariable(t.text);
  }
  throw std::runtime_error("Parse error: unexpected token '" + t.text + "'.");
}

std::unique_ptr<AbstractValue> CircuitRuntime::binary(const std::string &op, std::unique_ptr<AbstractValue> lhs,
                                                      std::unique_ptr<AbstractValue> rhs) {
  const bool lsec = isCiphertext(lhs.get()), rsec = isCiphertext(rhs.get());
  const bool commutative = (op == "+" || op == "+++" || op == "*" || op == "***" || op == "==" || op == "!=");
  if (lsec != rsec && commutative && rsec) std::swap(lhs, rhs);  // ciphertext becomes the receiver
  if (op == "+" || op == "+++") {
    lhs->add_inplace(*rhs);
  } else if (op == "-" || op == "---") {
    if (!lsec && rsec) {
      // public - secret: encrypt the public operand, then subtract (upstream discards this result)
      auto c = factory.createCiphertext(std::move(lhs));
      c->subtractInplace(*dynamic_cast<AbstractCiphertext *>(rhs.get()));
      return c;
    }
    lhs->subtract_inplace(*rhs);
  } else if (op == "*" || op == "***") {
    lhs->multiply_inplace(*rhs);
  } else if (op == "/") {
    lhs->divide_inplace(*rhs);
  } else if (op == "%") {
    lhs->modulo_inplace(*rhs);
  } else {
    if (op == "<") lhs->logicalLess_inplace(*rhs);
    else if (op == "<=") lhs->logicalLessEqual_inplace(*rhs);
    else if (op == ">") lhs->logicalGreater_inplace(*rhs);
    else if (op == ">=") lhs->logicalGreaterEqual_inplace(*rhs);
    else if (op == "==") lhs->logicalEqual_inplace(*rhs);
    else if (op == "!=") lhs->logicalNotEqual_inplace(*rhs);
    else throw std::runtime_error("Unknown binary operator encountered. Cannot continue!");
    // relational result becomes a Cleartext<bool> (RuntimeVisitor.cpp:103-107)
    if (!dynamic_cast<Cleartext<bool> *>(lhs.get())) return std::make_unique<Cleartext<bool>>(std::move(lhs));
  }
  return lhs;
}

OutputIdentifierValuePairs CircuitRuntime::getOutput(const std::string &outputs) {
  // each statement: `<id> = <var>;` (clone) or `<id> = <var>[<int>];` (rotateRows) -- RuntimeVisitor.cpp:489-530
  load(outputs);
  OutputIdentifierValuePairs result;
  while (peek().kind != Token::End) {
    const Token target = next();
    if (target.kind != Token::Ident || !accept("="))
      throw std::runtime_error("Block statements of given (in-/out)put AST must be of type Assignment. ");
    const Token src = next();
    if (src.kind != Token::Ident)
      throw std::runtime_error("Right-hand side of output AST is neither a Variable nor IndexAccess (e.g., y = __input0__ or y = __input0__[2]).");
    auto it = vars.find(src.text);
    if (it == vars.end()) throw std::runtime_error("Identifier '" + src.text + "' cannot be resolved.");
    std::unique_ptr<AbstractValue> value;
    if (accept("[")) {
      const Token idx = next();
      if (idx.kind != Token::Int || !accept("]"))
        throw std::runtime_error("Nested index accesses in right-hand side of output AST not allowed (e.g., y = __input0__[a[2]]).");
      if (!it->second.secret) throw std::runtime_error("Identifier '" + src.text + "' is not a ciphertext.");
      value = it->second.ctxt->rotateRows((int)idx.value);
    } else if (it->second.secret) {
      value = it->second.ctxt->clone();
    } else {
      value = it->second.clear->clone();
    }
    expect(";");
    result.emplace_back(target.text, std::move(value));
  }
  return result;
}

void CircuitRuntime::printOutput(const std::string &outputs, std::ostream &target) {
  for (const auto &v : getOutput(outputs)) {
    target << v.first << ": ";
    if (auto c = dynamic_cast<AbstractCiphertext *>(v.second.get())) target << factory.getString(*c) << std::endl;
    else if (auto p = dynamic_cast<ICleartext *>(v.second.get())) target << p->toString() << std::endl;
  }
}

// ---- recorded circuits ----
std::map<std::string, CircuitRuntime::Variable> CircuitRuntime::snapshot() const {
  std::map<std::string, Variable> copy;
  for (const auto &kv : vars) {
    Variable v;
    v.secret = kv.second.secret;
    if (kv.second.ctxt) v.ctxt = kv.second.ctxt->clone();  // copy-on-write reference
    if (kv.second.clear) v.clear = kv.second.clear->clone();
    copy.emplace(kv.first, std::move(v));
  }
  return copy;
}

CircuitRuntime::~CircuitRuntime() {
  if (graph)
    if (auto g = dynamic_cast<const GraphCapable *>(&factory)) {
      try { g->graphDestroy(graph); } catch (...) {}
    }
}

void CircuitRuntime::compile(const std::string &program) {
  auto g = dynamic_cast<const GraphCapable *>(&factory);
  if (!g) throw std::runtime_error("compile: this ciphertext factory cannot record circuits (no GraphCapable)");
  if (graph) throw std::runtime_error("compile: a circuit has already been recorded by this runtime");
  auto before = snapshot();
  executeAst(program);  // eager warm-up
  g->synchronize();
  vars = std::move(before);
  for (const auto &kv : vars)
    if (kv.second.ctxt) graphInputs[kv.first] = kv.second.ctxt->clone();
  g->graphBegin();
  try {
    executeAst(program);
  } catch (const std::exception &e) {
    g->graphAbort();
    throw std::runtime_error(std::string("while recording the circuit (only device work can be recorded; encryption, i.e. a "
                                         "secret declaration with a public initialiser, cannot): ") + e.what());
  } catch (...) {
    g->graphAbort();
    throw;
  }
  graph = g->graphEnd();
  g->graphLaunch(graph);
}

void CircuitRuntime::replay() {
  auto g = dynamic_cast<const GraphCapable *>(&factory);
  if (!g || !graph) throw std::runtime_error("replay: no recorded circuit");
  g->graphLaunch(graph);
}

void CircuitRuntime::setInput(const std::string &name, const std::vector<int64_t> &values) {
  auto g = dynamic_cast<const GraphCapable *>(&factory);
  auto it = graphInputs.find(name);
  if (!g || it == graphInputs.end()) throw std::runtime_error("setInput: '" + name + "' is not an input of a recorded circuit");
  g->rewriteCiphertext(*it->second, values);
}
void CircuitRuntime::setInputBatch(const std::string &name, const std::vector<std::vector<int64_t>> &perInstance) {
  auto g = dynamic_cast<const GraphCapable *>(&factory);
  auto it = graphInputs.find(name);
  if (!g || it == graphInputs.end()) throw std::runtime_error("setInputBatch: '" + name + "' is not an input of a recorded circuit");
  g->rewriteCiphertextBatch(*it->second, perInstance);
}
