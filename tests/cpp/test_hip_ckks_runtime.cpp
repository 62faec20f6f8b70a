// CKKS behind the plugin surface (HipCiphertextFactory with a CKKS HipSchemeConfig, driven directly and through the
// CircuitRuntime interpreter in batch mode), checked against the CPU oracle at RESIDUE level: the ciphertexts the plugin
// produced are downloaded, the oracle (test infrastructure, oracle/oracle.h) repeats the operation sequence on them with
// the same keys (same seeded sampling spec), and every word must agree; decoded slot values are checked to 1e-4 besides.
// The reference has no CKKS code or test: parity vs the reference is unpinned for everything here.
//   mode "cpu": encoder only (no GPU);  mode "gpu" (default): everything
#include <cmath>
#include <cstring>
#include <random>
#include <sstream>

#include "../../include/abc_hip.h"
#include "../../oracle/oracle.h"
#include "CircuitRuntime.hpp"
#include "CkksEncoder.hpp"
#include "HipCiphertext.hpp"
#include "HipCiphertextFactory.hpp"
#include "mini_test.hpp"

static std::vector<uint64_t> download(HipCiphertextFactory &f, const HipCiphertext &c) {
  std::vector<uint64_t> h(f.ciphertextWords(c.level()));
  abcHipCheck(abc_hip_memcpy_d2h(f.context(), h.data(), c.devicePtr(), h.size() * 8), "download");
  return h;
}
static const HipCiphertext &hip(const AbstractCiphertext &c) { return dynamic_cast<const HipCiphertext &>(c); }
static void expectSameWords(const std::vector<uint64_t> &got, const std::vector<uint64_t> &want, const char *what) {
  if (got.size() != want.size()) throw std::runtime_error(std::string(what) + ": size differs");
  for (size_t i = 0; i < got.size(); ++i)
    if (got[i] != want[i])
      throw std::runtime_error(std::string(what) + ": word " + std::to_string(i) + " got " + std::to_string(got[i]) + " want " +
                               std::to_string(want[i]));
}
static void expectClose(const std::vector<double> &got, const std::vector<double> &want, double tol, const char *what) {
  for (size_t i = 0; i < want.size(); ++i)
    if (!(std::fabs(got.at(i) - want[i]) <= tol * std::fmax(1.0, std::fabs(want[i]))))
      throw std::runtime_error(std::string(what) + ": slot " + std::to_string(i) + " got " + std::to_string(got[i]) + " want " +
                               std::to_string(want[i]));
}

struct Oracle {
  orc_ctx *c = nullptr;
  size_t n;
  int L;
  Oracle(int logn, const std::vector<int> &bits, uint64_t seed) : n((size_t)1 << logn), L((int)bits.size() - 1) {
    std::vector<uint64_t> primes(bits.size());
    if (orc_create_primes(n, bits.data(), (int)bits.size(), primes.data())) throw std::runtime_error("oracle primes");
    c = orc_ctx_create(2, logn, primes.data(), (int)primes.size(), 0);
    if (!c || orc_keygen(c, seed)) throw std::runtime_error("oracle context");
  }
  ~Oracle() { orc_ctx_destroy(c); }
  std::vector<uint64_t> mulRelinRescale(const std::vector<uint64_t> &a, const std::vector<uint64_t> &b, int nl) {
    std::vector<uint64_t> m(2 * nl * n), r(2 * (nl - 1) * n);
    if (orc_ckks_mul_relin(c, a.data(), b.data(), nl, m.data()) || orc_ckks_rescale(c, m.data(), 2, nl, r.data()))
      throw std::runtime_error("oracle mul");
    return r;
  }
  std::vector<uint64_t> rotate(const std::vector<uint64_t> &a, int nl, int steps) {
    std::vector<uint64_t> r(a.size());
    if (orc_rotate(c, a.data(), nl, steps, r.data())) throw std::runtime_error("oracle rotate");
    return r;
  }
  std::vector<uint64_t> add(const std::vector<uint64_t> &a, const std::vector<uint64_t> &b, int nl) {
    std::vector<uint64_t> r(a.size());
    orc_add(c, a.data(), b.data(), 2, nl, r.data());
    return r;
  }
};

static void encoderTests(MiniTest &t) {
  t.run("encoder round trip and agreement with the oracle's decoder", [] {
    const int logn = 12;
    const size_t n = 1 << logn;
    const std::vector<int> bits = {50, 40, 40, 50};
    Oracle o(logn, bits, 1);
    std::vector<uint64_t> primes;
    for (int j = 0; j < 3; j++) primes.push_back(orc_ctx_prime(o.c, j));
    CkksEncoder enc(n, primes);
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> dist(-100.0, 100.0);
    std::vector<double> v(n / 2);
    for (auto &x : v) x = dist(rng);
    const double scale = std::ldexp(1.0, 40);
    for (int nl = 3; nl >= 1; --nl) {
      std::vector<uint64_t> res((size_t)nl * n);
      enc.encode(v, scale, nl, res.data());
      std::vector<double> back;
      enc.decode(res.data(), nl, scale, back);
      expectClose(back, v, 1e-7, "round trip");
      // the oracle decodes plaintexts in NTT form
      std::vector<uint64_t> ntt(res);
      for (int j = 0; j < nl; j++) orc_ntt_forward(o.c, j, ntt.data() + (size_t)j * n);
      std::vector<double> re(n / 2), im(n / 2);
      if (orc_ckks_decode(o.c, ntt.data(), nl, scale, re.data(), im.data())) throw std::runtime_error("oracle decode");
      expectClose(re, v, 1e-7, "oracle decode of our encoding");
      // and the other way round
      std::vector<uint64_t> theirs((size_t)nl * n);
      std::vector<double> zero(n / 2, 0.0);
      if (orc_ckks_encode(o.c, v.data(), zero.data(), n / 2, scale, nl, theirs.data())) throw std::runtime_error("oracle encode");
      for (int j = 0; j < nl; j++) orc_ntt_inverse(o.c, j, theirs.data() + (size_t)j * n);
      enc.decode(theirs.data(), nl, scale, back);
      expectClose(back, v, 1e-7, "our decode of the oracle's encoding");
    }
  });
}

static std::string listOf(const std::vector<int> &v) {
  std::ostringstream os;
  os << "{";
  for (size_t i = 0; i < v.size(); ++i) os << (i ? ", " : "") << v[i];
  os << "}";
  return os.str();
}

int main(int argc, char **argv) {
  MiniTest t;
  encoderTests(t);
  if (argc > 1 && std::string(argv[1]) == "cpu") return t.summary();

  const uint64_t seed = 0xABC00C55ull;
  // ---- config 3 parameters: N = 2^14, {50,40,40,40 | 50}, scale 2^40; then the same on a SEAL-typical {60,40,40,40 | 60} chain
  //      (60-bit primes: integer kernels for their limbs, fp64 ones for the limbs between -- the plugin classes see no difference)
  const std::vector<std::vector<int>> chains = {{50, 40, 40, 40, 50}, {60, 40, 40, 40, 60}};
  for (const auto &bits : chains) {
    const std::string tag = bits[0] == 50 ? "" : " [60-bit chain]";
    HipSchemeConfig cfg;
    cfg.ckks = true;
    cfg.ringDegree = 16384;
    cfg.seed = seed;
    cfg.ckksBits = bits;
    HipCiphertextFactory f(cfg);
    Oracle o(14, cfg.ckksBits, seed);
    const size_t slots = 8192;
    std::mt19937_64 rng(3);
    std::uniform_real_distribution<double> dist(-1.0, 1.0);
    std::vector<double> x(slots), y(slots);
    for (auto &v : x) v = dist(rng);
    for (auto &v : y) v = dist(rng);

    t.run(("CKKS create / decrypt, add, subtract" + tag).c_str(), [&] {
      auto a = f.createCiphertext(x), b = f.createCiphertext(y);
      EXPECT_TRUE(hip(*a).level() == 4);
      std::vector<double> got, want(slots);
      f.decryptCiphertextReal(*a, got);
      expectClose(got, x, 1e-6, "decrypt");
      for (size_t i = 0; i < slots; ++i) want[i] = x[i] + y[i];
      auto s = a->add(*b);
      expectSameWords(download(f, hip(*s)), o.add(download(f, hip(*a)), download(f, hip(*b)), 4), "add vs oracle");
      f.decryptCiphertextReal(*s, got);
      expectClose(got, want, 1e-6, "add");
      for (size_t i = 0; i < slots; ++i) want[i] = x[i] - y[i];
      f.decryptCiphertextReal(*a->subtract(*b), got);
      expectClose(got, want, 1e-6, "subtract");
    });
    t.run(("CKKS multiply = mul + relinearise + rescale, residues equal to the oracle's at every level" + tag).c_str(), [&] {
      auto a = f.createCiphertext(x), b = f.createCiphertext(y);
      std::vector<double> want(x);
      for (int level = 4; level >= 2; --level) {
        EXPECT_TRUE(hip(*a).level() == level);
        const auto ra = download(f, hip(*a)), rb = download(f, hip(*b));
        a->multiplyInplace(*b);  // b stays at the top level: brought down to a's level inside
        EXPECT_TRUE(hip(*a).level() == level - 1);
        std::vector<uint64_t> rbl = rb;  // oracle: the operand at a's level
        for (int l = 4; l > level; --l) {
          std::vector<uint64_t> d(2 * (size_t)(l - 1) * o.n);
          orc_ckks_mod_switch(o.c, rbl.data(), 2, l, d.data());
          rbl.swap(d);
        }
        expectSameWords(download(f, hip(*a)), o.mulRelinRescale(ra, rbl, level), "multiply vs oracle");
        for (size_t i = 0; i < slots; ++i) want[i] *= y[i];
        std::vector<double> got;
        f.decryptCiphertextReal(*a, got);
        expectClose(got, want, 1e-4, "product");
      }
      EXPECT_TRUE(hip(*b).level() == 4);  // operand untouched
    });
    t.run(("CKKS plain operands and rotation" + tag).c_str(), [&] {
      auto a = f.createCiphertext(x);
      std::vector<double> got, want(slots);
      Cleartext<double> half(std::vector<double>{0.5});
      Cleartext<int> three(std::vector<int>{3});
      for (size_t i = 0; i < slots; ++i) want[i] = x[i] + 0.5;
      f.decryptCiphertextReal(*a->addPlain(half), got);
      expectClose(got, want, 1e-6, "addPlain");
      for (size_t i = 0; i < slots; ++i) want[i] = x[i] - 3;
      f.decryptCiphertextReal(*a->subtractPlain(three), got);
      expectClose(got, want, 1e-6, "subtractPlain");
      auto m = a->multiplyPlain(three);
      EXPECT_TRUE(hip(*m).level() == 3);
      for (size_t i = 0; i < slots; ++i) want[i] = 3 * x[i];
      f.decryptCiphertextReal(*m, got);
      expectClose(got, want, 1e-5, "multiplyPlain");
      auto neg = a->multiplyPlain(Cleartext<int>(std::vector<int>{-1}));
      EXPECT_TRUE(hip(*neg).level() == 4);  // negation spends no level
      for (size_t i = 0; i < slots; ++i) want[i] = -x[i];
      f.decryptCiphertextReal(*neg, got);
      expectClose(got, want, 1e-6, "negate");
      auto r = a->rotateRows(5);
      expectSameWords(download(f, hip(*r)), o.rotate(download(f, hip(*a)), 4, 5), "rotate vs oracle");
      for (size_t i = 0; i < slots; ++i) want[i] = x[(i + 5) % slots];
      f.decryptCiphertextReal(*r, got);
      expectClose(got, want, 1e-6, "rotateRows");
      // a sum of values at different levels: the higher one comes down
      auto s = m->add(*a->multiplyPlain(Cleartext<double>(std::vector<double>{0.25})));
      for (size_t i = 0; i < slots; ++i) want[i] = 3.25 * x[i];
      f.decryptCiphertextReal(*s, got);
      expectClose(got, want, 1e-5, "sum of two products");
      EXPECT_THROWS(a->addPlain(Cleartext<bool>(std::vector<bool>{true})));
    });
  }
  // ---- config 3 through the interpreter, batch mode: dot product of two length-8192 vectors (multiply, rescale,
  //      13 x rotate-and-add: VectorizerTest.cpp:169-173,209-214), B = 3 independent input pairs in one pass ----
  t.run("config 3: batched dot-product circuit through CircuitRuntime on the CKKS factory", [&] {
    HipSchemeConfig cfg;
    cfg.ckks = true;
    cfg.ringDegree = 16384;
    cfg.seed = seed;
    cfg.batch = 3;
    HipCiphertextFactory f(cfg);
    Oracle o(14, cfg.ckksBits, seed);
    const size_t slots = 8192;
    std::mt19937 rng(11);
    std::uniform_int_distribution<int> dist(-9, 9);
    std::vector<std::vector<int64_t>> xs(3, std::vector<int64_t>(slots)), ys(xs);
    std::vector<double> dots(3, 0.0);
    for (int b = 0; b < 3; ++b)
      for (size_t i = 0; i < slots; ++i) {
        xs[b][i] = dist(rng);
        ys[b][i] = dist(rng);
        dots[b] += (double)(xs[b][i] * ys[b][i]);
      }
    f.queueBatchedInput(xs);
    f.queueBatchedInput(ys);
    CircuitRuntime rt(f, "secret int __input0__ = {0}; secret int __input1__ = {0};");
    std::string prog = "secret int r = __input0__ *** __input1__;\n";
    for (int step = 4096; step >= 1; step /= 2) prog += "r = r +++ rotate(r, " + std::to_string(step) + ");\n";
    rt.executeAst(prog);
    auto out = rt.getOutput("y = r;");
    auto &res = *dynamic_cast<AbstractCiphertext *>(out[0].second.get());
    EXPECT_TRUE(hip(res).level() == 3);
    std::vector<std::vector<double>> dec;
    f.decryptCiphertextRealBatch(res, dec);
    for (int b = 0; b < 3; ++b)
      for (size_t i = 0; i < slots; i += 1021)
        if (std::fabs(dec[b][i] - dots[b]) > 1e-3 * std::fmax(1.0, std::fabs(dots[b])))
          throw std::runtime_error("instance " + std::to_string(b) + " slot " + std::to_string(i) + ": got " + std::to_string(dec[b][i]) +
                                   " want " + std::to_string(dots[b]));
    // the interface's integer view (decryptCiphertext rounds): instance 0
    std::vector<int64_t> ints;
    f.decryptCiphertext(res, ints);
    EXPECT_TRUE(ints[0] == (int64_t)std::llrint(dots[0]));
  });
  // ---- config 4: N = 2^15, 8x8 box sum on a 64x64 image (rotations 1,2,4 then 64,128,256, each followed by add) ----
  t.run("config 4: box-sum circuit through CircuitRuntime on a CKKS N = 32768 factory", [&] {
    HipSchemeConfig cfg;
    cfg.ckks = true;
    cfg.ringDegree = 32768;
    cfg.ckksBits = {40, 30, 30, 40};
    cfg.ckksScale = std::ldexp(1.0, 30);
    cfg.seed = seed;
    cfg.batch = 2;
    HipCiphertextFactory f(cfg);
    const int S = 64;
    std::mt19937 rng(5);
    std::uniform_int_distribution<int> dist(0, 255);
    std::vector<std::vector<int64_t>> imgs(2, std::vector<int64_t>(S * S));
    for (auto &im : imgs)
      for (auto &p : im) p = dist(rng);
    f.queueBatchedInput(imgs);
    CircuitRuntime rt(f, "secret int img = {0};");
    std::string prog = "secret int acc = img;\n";
    for (int r : {1, 2, 4, 64, 128, 256}) prog += "acc = acc +++ rotate(acc, " + std::to_string(r) + ");\n";
    rt.executeAst(prog);
    auto out = rt.getOutput("y = acc;");
    auto &res = *dynamic_cast<AbstractCiphertext *>(out[0].second.get());
    std::vector<std::vector<double>> dec;
    f.decryptCiphertextRealBatch(res, dec);
    const size_t slots = 16384;
    for (int b = 0; b < 2; ++b) {
      // slot p of the result = sum over the 8x8 window (dx, dy) of padded[p + dx + 64 dy]; pad-with-last beyond the image
      auto at = [&](size_t p) { return (double)(p < (size_t)S * S ? imgs[b][p] : imgs[b].back()); };
      for (size_t p = 0; p < (size_t)S * S; p += 97) {
        double want = 0;
        for (int dy = 0; dy < 8; ++dy)
          for (int dx = 0; dx < 8; ++dx) want += at((p + dx + 64 * dy) % slots);
        if (std::fabs(dec[b][p] - want) > 0.05)
          throw std::runtime_error("image " + std::to_string(b) + " pixel " + std::to_string(p) + ": got " + std::to_string(dec[b][p]) +
                                   " want " + std::to_string(want));
      }
    }
  });
  t.run("BFV factory refuses real inputs, CKKS factory exhausts its chain gracefully", [&] {
    HipCiphertextFactory bfv(4096, 0, 5);
    EXPECT_THROWS(bfv.createCiphertext(std::vector<double>{0.5}));
    HipSchemeConfig cfg;
    cfg.ckks = true;
    cfg.ringDegree = 4096;
    cfg.ckksBits = {40, 30, 40};
    cfg.ckksScale = std::ldexp(1.0, 30);
    cfg.seed = 9;
    HipCiphertextFactory f(cfg);
    auto a = f.createCiphertext(std::vector<double>{1.5, -2.0});
    a->multiplyInplace(*a);  // level 2 -> 1
    EXPECT_TRUE(hip(*a).level() == 1);
    std::vector<double> got;
    f.decryptCiphertextReal(*a, got);
    expectClose(got, {2.25, 4.0}, 1e-3, "square");
  });
  return t.summary();
}
