// HIP backend behind the plugin surface: the reference's SEAL-backed tests re-expressed against
// HipCiphertextFactory (N = 4096 like test/runtime/SealCiphertextFactoryTest.cpp:14,19 and
// test/runtime/RuntimeVisitorTest.cpp:16).  Every expected vector below is the reference's own.
#include "CircuitRuntime.hpp"
#include "HipCiphertext.hpp"
#include "HipCiphertextFactory.hpp"
#include "mini_test.hpp"

static const int N = 4096;
static std::vector<int64_t> dec(HipCiphertextFactory &f, AbstractCiphertext &c) {
  std::vector<int64_t> v;
  f.decryptCiphertext(c, v);
  return v;
}
// checkCiphertextData, SealCiphertextFactoryTest.cpp:22-41
static void checkPadded(HipCiphertextFactory &f, AbstractCiphertext &c, const std::vector<int64_t> &expected) {
  auto r = dec(f, c);
  EXPECT_TRUE(r.size() == (size_t)N);
  expectPrefix(r, expected);
  for (size_t i = expected.size(); i < r.size(); ++i) EXPECT_TRUE(r[i] == expected.back());
}
static std::vector<int64_t> outputOf(HipCiphertextFactory &f, OutputIdentifierValuePairs &out, const std::string &id) {
  for (auto &p : out)
    if (p.first == id) return dec(f, *dynamic_cast<AbstractCiphertext *>(p.second.get()));
  throw std::runtime_error("no output named " + id);
}

int main() {
  MiniTest t;
  HipCiphertextFactory f(N, 0, 0xABC00001ull);
  const std::vector<int64_t> d1 = {3, 3, 1, 4, 5, 9}, d2 = {0, 1, 2, 1, 10, 21};
  const std::vector<int64_t> rot = {123456, 3, 1, 4, 5, 9, 5, 2, 1, 5};
  Cleartext<int> p2(std::vector<int>{0, 1, 2, 1, 10, 21});

  t.run("createCiphertext", [&] { auto c = f.createCiphertext(d1); checkPadded(f, *c, d1); });
  t.run("add sub multiply (returning)", [&] {
    auto a = f.createCiphertext(d1), b = f.createCiphertext(d2);
    checkPadded(f, *a->add(*b), {3, 4, 3, 5, 15, 30});
    checkPadded(f, *a->subtract(*b), {3, 2, -1, 3, -5, -12});
    checkPadded(f, *a->multiply(*b), {0, 3, 2, 4, 50, 189});
    checkPadded(f, *a, d1);
    checkPadded(f, *b, d2);
  });
  t.run("add sub multiply (in place)", [&] {
    auto b = f.createCiphertext(d2);
    auto a = f.createCiphertext(d1); a->addInplace(*b); checkPadded(f, *a, {3, 4, 3, 5, 15, 30});
    a = f.createCiphertext(d1); a->subtractInplace(*b); checkPadded(f, *a, {3, 2, -1, 3, -5, -12});
    a = f.createCiphertext(d1); a->multiplyInplace(*b); checkPadded(f, *a, {0, 3, 2, 4, 50, 189});
  });
  t.run("plain operations (returning and in place)", [&] {
    auto a = f.createCiphertext(d1);
    checkPadded(f, *a->addPlain(p2), {3, 4, 3, 5, 15, 30});
    checkPadded(f, *a->subtractPlain(p2), {3, 2, -1, 3, -5, -12});
    checkPadded(f, *a->multiplyPlain(p2), {0, 3, 2, 4, 50, 189});
    checkPadded(f, *a, d1);
    a->multiplyPlainInplace(p2); checkPadded(f, *a, {0, 3, 2, 4, 50, 189});
    auto m = f.createCiphertext(d1);
    m->multiplyPlainInplace(Cleartext<int>(std::vector<int>{-1}));  // negate shortcut
    checkPadded(f, *m, {-3, -3, -1, -4, -5, -9});
    EXPECT_THROWS(a->addPlain(Cleartext<bool>(std::vector<bool>{true})));
  });
  t.run("rotateRows +4 / -24 / in place, wrap at N/2", [&] {
    auto c = f.createCiphertext(rot);
    const size_t row = N / 2, n0 = rot.size();
    auto check4 = [&](const std::vector<int64_t> &dv) {
      const size_t steps = 4;
      for (size_t i = 0; i < dv.size(); ++i) {
        if (i < std::min(n0 - steps, row - steps)) EXPECT_TRUE(dv[i] == rot[i + steps]);
        else if (i >= row - steps && i < row) EXPECT_TRUE(dv[i] == rot[i - (row - steps)]);
        else EXPECT_TRUE(dv[i] == rot[n0 - 1]);
      }
    };
    check4(dec(f, *c->rotateRows(4)));
    checkPadded(f, *c, rot);  // operand unchanged
    auto dv = dec(f, *c->rotateRows(-24));
    for (size_t i = 0; i < dv.size(); ++i) {
      if (i < 24 || i >= 24 + n0) EXPECT_TRUE(dv[i] == rot[n0 - 1]);
      else EXPECT_TRUE(dv[i] == rot[i - 24]);
    }
    c->rotateRowsInplace(4);
    check4(dec(f, *c));
  });
  t.run("unsupported operators throw, clone is deep, factories do not mix", [&] {
    auto a = f.createCiphertext(d1), b = f.createCiphertext(d2);
    EXPECT_THROWS(a->divide_inplace(*b));
    EXPECT_THROWS(a->logicalNot_inplace());
    EXPECT_THROWS(a->bitwiseXor_inplace(*b));
    auto c = a->clone();
    a->addInplace(*b);
    checkPadded(f, *c, d1);
    EXPECT_THROWS(f.createCiphertext(std::vector<int64_t>(N + 1, 1)));
    EXPECT_TRUE(f.getString(*c).substr(0, 14) == "[ 3,  3,  1,  ");
  });

  // ---- RuntimeVisitorTest.cpp programs ----
  const std::string in0 = "secret int __input0__ = {43, 1, 1, 1, 22, 11, 425, 0, 1, 7};";
  t.run("testRotateNegative", [&] {
    CircuitRuntime rt(f, in0);
    rt.executeAst("__input0__ = rotate(__input0__, -4);");
    auto out = rt.getOutput("y = __input0__;");
    expectPrefix(outputOf(f, out, "y"), {7, 7, 7, 7, 43, 1, 1, 1, 22, 11, 425, 0, 1, 7});
  });
  t.run("testRotatePositive", [&] {
    CircuitRuntime rt(f, in0);
    rt.executeAst("__input0__ = rotate(__input0__, 6);");
    auto out = rt.getOutput("y = __input0__;");
    expectPrefix(outputOf(f, out, "y"), {425, 0, 1, 7, 7, 7, 7, 7, 7});
  });
  t.run("testBinaryExpressionCtxtCtxt", [&] {
    CircuitRuntime rt(f, in0 + "secret int __input1__ = {24, 34, 222, 4, 1, 4, 9, 22, 1, 3};");
    rt.executeAst("secret int result = __input0__ *** __input1__; return result;");
    auto out = rt.getOutput("y = result;");
    expectPrefix(outputOf(f, out, "y"), {1032, 34, 222, 4, 22, 44, 3825, 0, 1, 21});
  });
  t.run("testBinaryExpressionCtxtPlaintext / PlaintextCtxt", [&] {
    for (const char *prog : {"int i = 19; secret int result = __input0__ *** i; return result;",
                             "int i = 19; secret int result = i *** __input0__; return result;"}) {
      CircuitRuntime rt(f, "secret int __input0__ = {43, 1, 1, 22, 11, 7};");
      rt.executeAst(prog);
      auto out = rt.getOutput("y = result; x = result[3];");
      expectPrefix(outputOf(f, out, "y"), {817, 19, 19, 418, 209, 133});
      expectPrefix(outputOf(f, out, "x"), {418});
    }
  });
  t.run("testForLoop: ten encrypted additions", [&] {
    CircuitRuntime rt(f, in0);
    // the reference's program text, test/runtime/RuntimeVisitorTest.cpp:557-564
    rt.executeAst("int LIMIT = 10; secret int result = 0; for (int i = 0; i < LIMIT; i = i + 1) { result = result + __input0__; } return;");
    auto out = rt.getOutput("y = result;");
    expectPrefix(outputOf(f, out, "y"), {430, 10, 10, 10, 220, 110, 4250, 0, 10, 70});
  });
  t.run("secret declaration round trip; public minus secret is computed correctly", [&] {
    CircuitRuntime rt(f, "secret int a = {5, 6, 7};");
    rt.executeAst("secret int b = {1, 2, 3}; secret int c = 10 --- a; secret int d = a --- b;");
    auto out = rt.getOutput("b = b; c = c; d = d;");
    expectPrefix(outputOf(f, out, "b"), {1, 2, 3});
    expectPrefix(outputOf(f, out, "c"), {5, 4, 3});
    expectPrefix(outputOf(f, out, "d"), {4, 4, 4});
  });
  t.run("testFullAssignmentToCiphertext", [&] {
    CircuitRuntime rt(f, "");
    rt.executeAst("secret int fixedKey = {3, 2, 1, 3, 4, 9, 11, 333, 22, 434, 3430, 2211}; return;");
    auto out = rt.getOutput("result = fixedKey;");
    expectPrefix(outputOf(f, out, "result"), {3, 2, 1, 3, 4, 9, 11, 333, 22, 434, 3430, 2211});
  });
  t.run("must-throw programs", [&] {
    CircuitRuntime rt(f, in0);
    EXPECT_THROWS(rt.executeAst("secret int r = __input0__ / __input0__;"));
    EXPECT_THROWS(rt.executeAst("secret int r = __input0__ < __input0__;"));
    EXPECT_THROWS(rt.executeAst("secret int r = rotate(__input0__ +++ __input0__, 2);"));
    EXPECT_THROWS(rt.executeAst("secret int r = rotate(__input0__, 5000);"));
  });
  // ---- batch mode: one pass of the unchanged interpreter evaluates the circuit on B independent input sets ----
  t.run("batch mode: 5 circuit instances in one interpreter pass", [&] {
    const size_t B = 5;
    HipCiphertextFactory fb(N, 0, 0xABC00002ull, B);
    std::vector<std::vector<int64_t>> x(B), y(B);
    for (size_t b = 0; b < B; ++b)
      for (int i = 0; i < 12; ++i) {
        x[b].push_back((int64_t)(7 * b + 3 * i + 1) % 40);
        y[b].push_back((int64_t)(5 * b + i * i + 2) % 40);
      }
    fb.queueBatchedInput(x);  // consumed by the declaration of __input0__
    fb.queueBatchedInput(y);  // ... of __input1__
    CircuitRuntime rt(fb, "secret int __input0__ = {0}; secret int __input1__ = {0}; int __input2__ = {2, 3, 4, 5, 6, 7, 8, 9, 1, 2, 3, 4};");
    rt.executeAst("secret int p = __input0__ *** __input1__; secret int r = rotate(p, 2); "
                  "secret int result = (r +++ __input0__) --- __input2__; return result;");  // one ct x ct level: N = 4096
    auto out = rt.getOutput("y = result;");
    std::vector<std::vector<int64_t>> got;
    for (auto &pr : out)
      if (pr.first == "y") fb.decryptCiphertextBatch(*dynamic_cast<AbstractCiphertext *>(pr.second.get()), got);
    EXPECT_TRUE(got.size() == B);
    const std::vector<int64_t> pub = {2, 3, 4, 5, 6, 7, 8, 9, 1, 2, 3, 4};
    for (size_t b = 0; b < B; ++b) {
      // slots are padded with the last value, rotate(p, 2) brings slot i+2 to slot i
      auto at = [&](const std::vector<int64_t> &v, size_t i) { return i < v.size() ? v[i] : v.back(); };
      for (size_t i = 0; i < 10; ++i) {
        const int64_t want = at(x[b], i + 2) * at(y[b], i + 2) + at(x[b], i) - pub[i];
        if (got[b][i] != want)
          throw std::runtime_error("instance " + std::to_string(b) + " slot " + std::to_string(i) + ": got " +
                                   std::to_string(got[b][i]) + " want " + std::to_string(want));
      }
    }
    // the reference's single-result view is instance 0
    std::vector<int64_t> first;
    for (auto &pr : out)
      if (pr.first == "y") fb.decryptCiphertext(*dynamic_cast<AbstractCiphertext *>(pr.second.get()), first);
    EXPECT_TRUE(first == got[0]);
    EXPECT_THROWS(fb.queueBatchedInput(std::vector<std::vector<int64_t>>(B + 1, std::vector<int64_t>{1})));
  });
  t.run("recorded circuit: compile once, replay on new inputs, same results as the eager interpreter", [&] {
    const std::string inputs = "secret int a = {3, 3, 1, 4, 5, 9}; secret int b = {0, 1, 2, 1, 10, 21}; int k = {2, 2, 2, 2, 2, 2};";
    const std::string program =
        "secret int r = a *** b;\n"
        "r = r +++ rotate(r, 1);\n"
        "for (int i = 0; i < 2; i = i + 1) { r = r +++ a; }\n"  // a public loop: unrolled into the recording
        "r = r --- (b *** k);\n"
        "a = a +++ b;\n";  // an input that is overwritten: the recording must keep reading the original buffer
    auto expect = [&](const std::vector<int64_t> &va, const std::vector<int64_t> &vb) {
      // padded semantics: slot i beyond the data holds the last value
      auto at = [](const std::vector<int64_t> &v, size_t i) { return i < v.size() ? v[i] : v.back(); };
      std::vector<int64_t> r(8);
      for (size_t i = 0; i < 8; ++i) {
        const int64_t m0 = at(va, i) * at(vb, i), m1 = at(va, i + 1) * at(vb, i + 1);
        r[i] = m0 + m1 + 2 * at(va, i) - 2 * at(vb, i);
      }
      return r;
    };
    CircuitRuntime rt(f, inputs);
    rt.compile(program);
    EXPECT_TRUE(rt.compiled());
    {
      auto out = rt.getOutput("y = r; z = a;");
      expectPrefix(outputOf(f, out, "y"), expect(d1, d2));
      expectPrefix(outputOf(f, out, "z"), {3, 4, 3, 5, 15, 30});
    }
    const std::vector<int64_t> a2 = {7, 1, 0, 2, 9, 4}, b2 = {5, 5, 1, 3, 2, 6};
    rt.setInput("a", a2);
    rt.setInput("b", b2);
    rt.replay();
    {
      auto out = rt.getOutput("y = r; z = a;");
      expectPrefix(outputOf(f, out, "y"), expect(a2, b2));
      expectPrefix(outputOf(f, out, "z"), {12, 6, 1, 5, 11, 10});
    }
    rt.replay();  // replays are idempotent: inputs are never written by the recording
    {
      auto out = rt.getOutput("y = r;");
      expectPrefix(outputOf(f, out, "y"), expect(a2, b2));
    }
    // other work on the same factory between replays must not disturb the recording's buffers
    auto other = f.createCiphertext(d1);
    for (int i = 0; i < 4; ++i) other = other->multiply(*other->rotateRows(1));
    rt.setInput("a", d1);
    rt.setInput("b", d2);
    rt.replay();
    {
      auto out = rt.getOutput("y = r;");
      expectPrefix(outputOf(f, out, "y"), expect(d1, d2));
    }
    // the recording multiplies by the cached plaintext of k: more distinct constants than the factory's plaintext cache holds,
    // encoded eagerly afterwards, evict that entry -- its block must stay parked for the graph, not be recycled under it
    {
      auto pile = f.createCiphertext(d1);
      for (int v = 0; v < 40; ++v) {
        pile->multiplyPlainInplace(Cleartext<int>(std::vector<int>{v + 3, 1, v + 5}));
        pile = f.createCiphertext(d1);  // fresh ciphertext every round: freed blocks of the plaintext's size get reused
      }
      rt.setInput("a", a2);
      rt.setInput("b", b2);
      rt.replay();
      auto out = rt.getOutput("y = r;");
      expectPrefix(outputOf(f, out, "y"), expect(a2, b2));
    }
    // a program that encrypts inside cannot be recorded: clean error, the factory stays usable
    CircuitRuntime bad(f, "secret int a = {1, 2};");
    EXPECT_THROWS(bad.compile("secret int t = {4, 5}; a = a +++ t;"));
    checkPadded(f, *f.createCiphertext(d1)->add(*f.createCiphertext(d2)), {3, 4, 3, 5, 15, 30});
  });
  return t.summary();
}
