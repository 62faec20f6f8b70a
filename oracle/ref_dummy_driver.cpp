// oracle/ref_dummy_driver.cpp -- TEST INFRASTRUCTURE.  Drives the REFERENCE's own Parser +
// TypeCheckingVisitor + RuntimeVisitor + DummyCiphertextFactory (compiled by oracle/build_ref.sh from
// /root/reference) on the config-1 programs and prints what the reference outputs.  The captured output
// is committed as tests/golden/ref_dummy_runtime.txt (a fixture: inputs and expected outputs only).
#include <iostream>
#include <string>

#include "ast_opt/parser/Parser.h"
#include "ast_opt/runtime/DummyCiphertextFactory.h"
#include "ast_opt/runtime/RuntimeVisitor.h"
#include "ast_opt/utilities/Scope.h"
#include "ast_opt/visitor/TypeCheckingVisitor.h"

static void run(const std::string &title, const std::string &inputs, const std::string &program, const std::string &outputs,
                const std::vector<std::string> &secretInputs) {
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
  DummyCiphertextFactory factory;
  RuntimeVisitor rv(factory, *astInput, taint);
  rv.executeAst(*astProgram);
  std::cout << "# " << title << std::endl;
  rv.printOutput(*astOutput, std::cout);
}

int main() {
  run("config1: element-wise ops on two length-4 secret vectors",
      "secret int __input0__ = {1, 2, 3, 4}; secret int __input1__ = {10, 20, 30, 40};",
      "secret int s = __input0__ +++ __input1__; secret int p = __input0__ *** __input1__; secret int d = __input1__ --- __input0__; return;",
      "y = s; p = p; d = d;", {"__input0__", "__input1__"});
  run("ct x public scalar needs equal sizes on the dummy backend: vector operand",
      "secret int __input0__ = {43, 1, 1, 22, 11, 7};",
      "int i = {19, 19, 19, 19, 19, 19}; secret int result = __input0__ *** i; return;",
      "y = result;", {"__input0__"});
  run("ten additions in a public for loop",
      "secret int __input0__ = {43, 1, 1, 1, 22, 11, 425, 0, 1, 7};",
      "secret int result = __input0__; for (int i = 0; i < 9; i = i + 1) { result = result +++ __input0__; } return;",
      "y = result;", {"__input0__"});
  return 0;
}
