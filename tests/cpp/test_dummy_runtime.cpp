// BASELINE.json config 1 -- "element-wise add of two length-4 secret vectors, dummy (cleartext) runtime on
// CPU": the host plumbing (CircuitRuntime + plugin surface) on the cleartext backend.  Known answers are
// the reference's own: test/runtime/DummyCiphertextFactoryTest.cpp:37-237 and the outputs of the
// reference's RuntimeVisitor + DummyCiphertextFactory on the same programs (tests/golden/ref_dummy_runtime.txt,
// produced by oracle/build_ref.sh from the reference sources).
#include <sstream>

#include "CircuitRuntime.hpp"
#include "DummyCiphertextFactory.hpp"
#include "mini_test.hpp"

static std::vector<int64_t> dec(DummyCiphertextFactory &f, AbstractCiphertext &c) {
  std::vector<int64_t> v;
  f.decryptCiphertext(c, v);
  return v;
}

int main(int argc, char **argv) {
  MiniTest t;
  DummyCiphertextFactory f;
  const std::vector<int64_t> d1 = {3, 3, 1, 4, 5, 9}, d2 = {0, 1, 2, 1, 10, 21};
  const std::vector<int> p2 = {0, 1, 2, 1, 10, 21};

  t.run("createCiphertext keeps the values, no padding", [&] {
    auto c = f.createCiphertext(d1);
    EXPECT_TRUE(dec(f, *c) == d1);
  });
  t.run("add / sub / multiply and in-place twins", [&] {
    auto a = f.createCiphertext(d1), b = f.createCiphertext(d2);
    EXPECT_TRUE(dec(f, *a->add(*b)) == (std::vector<int64_t>{3, 4, 3, 5, 15, 30}));
    EXPECT_TRUE(dec(f, *a->subtract(*b)) == (std::vector<int64_t>{3, 2, -1, 3, -5, -12}));
    EXPECT_TRUE(dec(f, *a->multiply(*b)) == (std::vector<int64_t>{0, 3, 2, 4, 50, 189}));
    EXPECT_TRUE(dec(f, *a) == d1);  // operands untouched
    a->addInplace(*b);
    EXPECT_TRUE(dec(f, *a) == (std::vector<int64_t>{3, 4, 3, 5, 15, 30}));
  });
  t.run("plain operations", [&] {
    auto a = f.createCiphertext(d1);
    Cleartext<int> pl(p2);
    EXPECT_TRUE(dec(f, *a->addPlain(pl)) == (std::vector<int64_t>{3, 4, 3, 5, 15, 30}));
    EXPECT_TRUE(dec(f, *a->subtractPlain(pl)) == (std::vector<int64_t>{3, 2, -1, 3, -5, -12}));
    EXPECT_TRUE(dec(f, *a->multiplyPlain(pl)) == (std::vector<int64_t>{0, 3, 2, 4, 50, 189}));
  });
  t.run("size mismatch and rotate throw", [&] {
    auto a = f.createCiphertext(d1), b = f.createCiphertext(std::vector<int64_t>{1, 2});
    EXPECT_THROWS(a->addInplace(*b));
    EXPECT_THROWS(a->rotateRows(3));
    EXPECT_THROWS(a->divide_inplace(*b));
  });
  std::ostringstream printed;
  t.run("config 1 through the interpreter", [&] {
    CircuitRuntime rt(f, "secret int __input0__ = {1, 2, 3, 4}; secret int __input1__ = {10, 20, 30, 40};");
    rt.executeAst(
        "secret int s = __input0__ +++ __input1__;"
        "secret int p = __input0__ *** __input1__;"
        "secret int d = __input1__ --- __input0__;");
    auto out = rt.getOutput("y = s; p = p; d = d;");
    EXPECT_TRUE(out.size() == 3);
    EXPECT_TRUE(dec(f, *dynamic_cast<AbstractCiphertext *>(out[0].second.get())) == (std::vector<int64_t>{11, 22, 33, 44}));
    EXPECT_TRUE(dec(f, *dynamic_cast<AbstractCiphertext *>(out[1].second.get())) == (std::vector<int64_t>{10, 40, 90, 160}));
    EXPECT_TRUE(dec(f, *dynamic_cast<AbstractCiphertext *>(out[2].second.get())) == (std::vector<int64_t>{9, 18, 27, 36}));
    rt.printOutput("y = s; p = p; d = d;", printed);
  });
  t.run("public arithmetic, loops and returns", [&] {
    CircuitRuntime rt(f, "int __input0__ = {43, 1, 1, 1, 22, 11, 425, 0, 1, 7};");
    rt.executeAst("int sum = 10 + 25; int acc = 0; for (int i = 0; i < 10; i = i + 1) { acc = acc + i; } return sum; int never = 1;");
    auto out = rt.getOutput("y = sum; z = acc;");
    EXPECT_TRUE(dynamic_cast<Cleartext<int> *>(out[0].second.get())->getData() == std::vector<int>{35});
    EXPECT_TRUE(dynamic_cast<Cleartext<int> *>(out[1].second.get())->getData() == std::vector<int>{45});
    EXPECT_THROWS(rt.getOutput("q = never;"));
  });
  t.run("unsupported constructs throw std::runtime_error", [&] {
    CircuitRuntime rt(f, "secret int a = {1, 2, 3, 4};");
    EXPECT_THROWS(rt.executeAst("secret int b = a / a;"));
    EXPECT_THROWS(rt.executeAst("secret int b = foo(a, 1);"));
    EXPECT_THROWS(rt.executeAst("secret int b = rotate(a +++ a, 1);"));
    EXPECT_THROWS(rt.executeAst("secret int c;"));
    EXPECT_THROWS(CircuitRuntime(f, "a = 3;"));
  });
  if (argc > 1 && std::string(argv[1]) == "--print") {
    // the three programs of oracle/ref_dummy_driver.cpp, printed in the reference driver's format so that the
    // output can be compared verbatim with tests/golden/ref_dummy_runtime.txt
    struct Case { const char *title, *inputs, *program, *outputs; };
    const Case cases[] = {
        {"config1: element-wise ops on two length-4 secret vectors",
         "secret int __input0__ = {1, 2, 3, 4}; secret int __input1__ = {10, 20, 30, 40};",
         "secret int s = __input0__ +++ __input1__; secret int p = __input0__ *** __input1__; secret int d = __input1__ --- __input0__; return;",
         "y = s; p = p; d = d;"},
        {"ct x public scalar needs equal sizes on the dummy backend: vector operand",
         "secret int __input0__ = {43, 1, 1, 22, 11, 7};",
         "int i = {19, 19, 19, 19, 19, 19}; secret int result = __input0__ *** i; return;", "y = result;"},
        {"ten additions in a public for loop", "secret int __input0__ = {43, 1, 1, 1, 22, 11, 425, 0, 1, 7};",
         "secret int result = __input0__; for (int i = 0; i < 9; i = i + 1) { result = result +++ __input0__; } return;",
         "y = result;"},
    };
    std::cout << "=== reference-format output ===" << std::endl;
    for (const auto &c : cases) {
      CircuitRuntime rt(f, c.inputs);
      rt.executeAst(c.program);
      std::cout << "# " << c.title << std::endl;
      rt.printOutput(c.outputs, std::cout);
    }
  }
  return t.summary();
}
