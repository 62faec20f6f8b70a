// oracle/ref_hip_dropin.cpp -- TEST INFRASTRUCTURE: the drop-in proof.  The REFERENCE's own Parser,
// TypeCheckingVisitor and RuntimeVisitor (compiled from /root/reference by oracle/build_ref.sh) drive this
// repository's HipCiphertextFactory through ABC's real AbstractCiphertext / AbstractCiphertextFactory headers
// (abc_amd/runtime compiled with -DABC_HIP_USE_REFERENCE_HEADERS).  Programs and expected slot values are
// those of test/runtime/RuntimeVisitorTest.cpp (:67-107, :195-222, :224-262, :264-342, :509-547, :549-594, :596-626).
// The binary lands in oracle/_ref/ and runs on the GPU box (tests/test_host_runtime.py, -m gpu).
#include <iostream>
#include <string>
#include <unordered_map>

#include "HipCiphertextFactory.hpp"
#include "ast_opt/parser/Parser.h"
#include "ast_opt/runtime/RuntimeVisitor.h"
#include "ast_opt/utilities/Scope.h"
#include "ast_opt/visitor/TypeCheckingVisitor.h"

static int failures = 0;

static void runCase(HipCiphertextFactory &factory, const std::string &title, const std::string &inputs, const std::string &program,
                    const std::string &outputs, const std::vector<std::string> &secretInputs,
                    const std::unordered_map<std::string, std::vector<int64_t>> &expected) {
  try {
    auto astInput = Parser::parse(inputs);
    auto astProgram = Parser::parse(program);
    auto astOutput = Parser::parse(outputs);
    TypeCheckingVisitor tcv;
    auto rootScope = std::make_unique<Scope>(*astProgram);
    for (const auto &id : secretInputs) {
      auto scoped = std::make_unique<ScopedIdentifier>(*rootScope, id);
      rootScope->addIdentifier(id);
      tcv.addVariableDatatype(*scoped, Datatype(Type::INT, true));
    }
    tcv.setRootScope(std::move(rootScope));
    astProgram->accept(tcv);
    auto taint = tcv.getSecretTaintedNodes();
    RuntimeVisitor rv(factory, *astInput, taint);
    rv.executeAst(*astProgram);
    auto result = rv.getOutput(*astOutput);
    bool ok = result.size() == expected.size();
    for (auto &[id, value] : result) {
      auto ctxt = dynamic_cast<AbstractCiphertext *>(value.get());
      std::vector<int64_t> plain;
      if (!ctxt || !expected.count(id)) { ok = false; continue; }
      factory.decryptCiphertext(*ctxt, plain);
      const auto &want = expected.at(id);
      for (size_t i = 0; i < want.size(); ++i) ok = ok && plain.at(i) == want[i];
    }
    std::cout << (ok ? "[  OK  ] " : "[ FAIL ] ") << title << std::endl;
    failures += !ok;
  } catch (const std::exception &e) {
    std::cout << "[ FAIL ] " << title << ": " << e.what() << std::endl;
    ++failures;
  }
}

int main() {
  HipCiphertextFactory factory(4096, 0, 0xABC00001ull);
  const std::string in0 = "secret int __input0__ = {43, 1, 1, 1, 22, 11, 425, 0, 1, 7};";
  runCase(factory, "testRotateNegative", in0, "__input0__ = rotate(__input0__, -4);", "y = __input0__;", {"__input0__"},
          {{"y", {7, 7, 7, 7, 43, 1, 1, 1, 22, 11, 425, 0, 1, 7}}});
  runCase(factory, "testRotatePositive", in0, "__input0__ = rotate(__input0__, 6);", "y = __input0__;", {"__input0__"},
          {{"y", {425, 0, 1, 7, 7, 7, 7, 7, 7}}});
  runCase(factory, "testBinaryExpressionCtxtCtxt", in0 + " secret int __input1__ = {24, 34, 222, 4, 1, 4, 9, 22, 1, 3};",
          "secret int result = __input0__ *** __input1__; return result;", "y = result;", {"__input0__", "__input1__"},
          {{"y", {1032, 34, 222, 4, 22, 44, 3825, 0, 1, 21}}});
  runCase(factory, "testBinaryExpressionCtxtPlaintext", "secret int __input0__ = {43, 1, 1, 22, 11, 7};",
          "int i = 19; secret int result = __input0__ *** i; return result;", "y = result; x = result[3];", {"__input0__"},
          {{"y", {817, 19, 19, 418, 209, 133}}, {"x", {418}}});
  runCase(factory, "testBinaryExpressionPlaintextCtxt", "secret int __input0__ = {43, 1, 1, 22, 11, 7};",
          "int i = 19; secret int result = i *** __input0__; return result;", "y = result; x = result[3];", {"__input0__"},
          {{"y", {817, 19, 19, 418, 209, 133}}, {"x", {418}}});
  runCase(factory, "testForLoop", in0,
          "int LIMIT = 10; secret int result = 0; for (int i = 0; i < LIMIT; i = i + 1) { result = result + __input0__; } return;",
          "y = result;", {"__input0__"}, {{"y", {430, 10, 10, 10, 220, 110, 4250, 0, 10, 70}}});
  runCase(factory, "testFullAssignmentToCiphertext", "",
          "secret int fixedKey = {3, 2, 1, 3, 4, 9, 11, 333, 22, 434, 3430, 2211}; return;", "result = fixedKey;", {},
          {{"result", {3, 2, 1, 3, 4, 9, 11, 333, 22, 434, 3430, 2211}}});
  runCase(factory, "secret declaration inside the program", "",
          "secret int sum = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10}; return sum;", "y = sum;", {},
          {{"y", {1, 2, 3, 4, 5, 6, 7, 8, 9, 10}}});
  // batch mode: the reference's interpreter, unchanged, evaluates testBinaryExpressionCtxtCtxt on four input sets at once
  try {
    const size_t B = 4;
    HipCiphertextFactory batched(4096, 0, 0xABC00004ull, B);
    std::vector<std::vector<int64_t>> x(B), y(B);
    for (size_t b = 0; b < B; ++b)
      for (int i = 0; i < 10; ++i) {
        x[b].push_back((int64_t)(11 * b + 5 * i + 3) % 97);
        y[b].push_back((int64_t)(7 * b + i * i + 1) % 89);
      }
    batched.queueBatchedInput(x);
    batched.queueBatchedInput(y);
    auto astInput = Parser::parse(std::string("secret int __input0__ = {0}; secret int __input1__ = {0};"));
    auto astProgram = Parser::parse(std::string("secret int result = __input0__ *** __input1__; return result;"));
    auto astOutput = Parser::parse(std::string("y = result;"));
    TypeCheckingVisitor tcv;
    auto rootScope = std::make_unique<Scope>(*astProgram);
    for (const std::string id : {"__input0__", "__input1__"}) {
      auto scoped = std::make_unique<ScopedIdentifier>(*rootScope, id);
      rootScope->addIdentifier(id);
      tcv.addVariableDatatype(*scoped, Datatype(Type::INT, true));
    }
    tcv.setRootScope(std::move(rootScope));
    astProgram->accept(tcv);
    auto taint = tcv.getSecretTaintedNodes();
    RuntimeVisitor rv(batched, *astInput, taint);
    rv.executeAst(*astProgram);
    auto result = rv.getOutput(*astOutput);
    bool ok = result.size() == 1;
    for (auto &[id, value] : result) {
      std::vector<std::vector<int64_t>> all;
      batched.decryptCiphertextBatch(*dynamic_cast<AbstractCiphertext *>(value.get()), all);
      ok = ok && all.size() == B;
      for (size_t b = 0; ok && b < B; ++b)
        for (size_t i = 0; i < 10; ++i) ok = ok && all[b][i] == x[b][i] * y[b][i];
    }
    std::cout << (ok ? "[  OK  ] " : "[ FAIL ] ") << "batch mode: four instances in one RuntimeVisitor pass" << std::endl;
    failures += !ok;
  } catch (const std::exception &e) {
    std::cout << "[ FAIL ] batch mode: " << e.what() << std::endl;
    ++failures;
  }
  std::cout << (failures ? "FAILED " : "passed ") << "reference RuntimeVisitor over HipCiphertextFactory, failures=" << failures
            << std::endl;
  return failures ? 1 : 0;
}
