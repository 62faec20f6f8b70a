// CircuitRuntime -- host-side interpreter for the circuit subset of ABC's input language that reaches the
// ciphertext plugin surface.  It is the counterpart of the reference's SpecialRuntimeVisitor
// (include/ast_opt/runtime/RuntimeVisitor.h:27-141, src/runtime/RuntimeVisitor.cpp) restricted to what the
// runtime tests exercise (test/runtime/RuntimeVisitorTest.cpp): vector declarations, `+ - *` and the FHE
// spellings `+++ --- ***`, relational operators on public values, `rotate(var, k)`, public `for` loops,
// `return`, and the output forms `y = v;` / `x = v[i];`.  It issues exactly the operation sequence the
// visitor issues against an AbstractCiphertextFactory:
//   * every variable read clones the stored value                         (RuntimeVisitor.cpp:429-443)
//   * commutative op with exactly one secret side: the ciphertext becomes the receiver (:60-64)
//   * secret declaration with a public initialiser -> factory.createCiphertext          (:409-416)
//   * rotate needs (variable, integer literal)                                          (:128-159)
//   * output `x = v[i]` is rotateRows(i) of the stored ciphertext                       (:511-517)
// Parsing ABC's full language (src/parser) is out of scope; programs are given in the same three source
// strings the reference tests use (inputs / program / outputs).
//
// Deviation, deliberate: `public - secret` is computed correctly (encrypt the public operand, then
// subtract).  Upstream's Cleartext<int>::subtract_inplace builds that result and discards it
// (include/ast_opt/runtime/Cleartext.h:349-360).
#pragma once

#include <map>
#include <string>
#include <utility>
#include <vector>

#include "plugin_api.hpp"

typedef std::vector<std::pair<std::string, std::unique_ptr<AbstractValue>>> OutputIdentifierValuePairs;

class CircuitRuntime {
 public:
  // `inputs`: a block of declarations, e.g. "secret int __input0__ = {43, 1, 1}; int __input1__ = {1, 2, 3};"
  CircuitRuntime(AbstractCiphertextFactory &factory, const std::string &inputs);
  void executeAst(const std::string &program);
  OutputIdentifierValuePairs getOutput(const std::string &outputs);
  void printOutput(const std::string &outputs, std::ostream &target);

  // ---- recorded circuits (needs a factory that implements GraphCapable, e.g. HipCiphertextFactory) ----
  // compile: interpret `program` once eagerly (fills plaintext caches, sizes scratch), roll the variables back, interpret it
  // again with the factory in recording mode, run the recording once.  Afterwards the variables hold the program's results,
  // as after executeAst -- and replay() re-evaluates the SAME operation sequence on whatever the input ciphertexts contain
  // now (setInput), without interpreting or dispatching anything on the host: one graph launch.
  // Static circuits only: the recorded sequence is the one this interpretation took (public loops are unrolled into it).
  // Value lifetimes need no analysis pass: a variable read is a copy-on-write reference (no device copy), an in-place
  // operation on a value nobody else holds runs in place, and buffers that die inside the recording are reused inside it.
  void compile(const std::string &program);
  void replay();
  bool compiled() const { return graph != nullptr; }
  // new contents for an input declared in the constructor's input block (same buffer, same address)
  void setInput(const std::string &name, const std::vector<int64_t> &values);
  void setInputBatch(const std::string &name, const std::vector<std::vector<int64_t>> &perInstance);
  ~CircuitRuntime();

  struct Token {
    enum Kind { End, Ident, Int, Punct } kind = End;
    std::string text;
    long value = 0;
  };

 private:
  struct Variable {
    bool secret = false;
    std::unique_ptr<AbstractCiphertext> ctxt;
    std::unique_ptr<ICleartext> clear;
  };
  struct ReturnReached {};

  AbstractCiphertextFactory &factory;
  std::map<std::string, Variable> vars;
  void *graph = nullptr;
  std::map<std::string, std::unique_ptr<AbstractCiphertext>> graphInputs;  // holds the recorded input buffers alive
  std::map<std::string, Variable> snapshot() const;
  std::vector<Token> toks;
  size_t pos = 0;

  void load(const std::string &src);
  const Token &peek(size_t ahead = 0) const;
  Token next();
  bool accept(const std::string &punct);
  void expect(const std::string &punct);
  void skipStatementOrBlock();

  void statement(bool inputsOnly);
  void block();
  void declaration(bool secret);
  void assignment();
  void forLoop();
  void store(const std::string &name, bool secretVar, std::unique_ptr<AbstractValue> value, bool declare);

  std::unique_ptr<AbstractValue> expression();
  std::unique_ptr<AbstractValue> relational();
  std::unique_ptr<AbstractValue> additive();
  std::unique_ptr<AbstractValue> multiplicative();
  std::unique_ptr<AbstractValue> primary();
  std::unique_ptr<AbstractValue> binary(const std::string &op, std::unique_ptr<AbstractValue> lhs,
                                        std::unique_ptr<AbstractValue> rhs);
  std::unique_ptr<AbstractValue> readVariable(const std::string &name);
};
