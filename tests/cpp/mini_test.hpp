// a few lines of test harness (GoogleTest is not installed in the image)
#pragma once
#include <cstdio>
#include <cstdint>
#include <functional>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

struct MiniTest {
  int failed = 0, passed = 0;
  void run(const char *name, const std::function<void()> &fn) {
    try {
      fn();
      ++passed;
      std::printf("[  OK  ] %s\n", name);
    } catch (const std::exception &e) {
      ++failed;
      std::printf("[ FAIL ] %s: %s\n", name, e.what());
    }
    std::fflush(stdout);
  }
  int summary() const {
    std::printf("%d passed, %d failed\n", passed, failed);
    return failed ? 1 : 0;
  }
};

inline std::string vecToString(const std::vector<int64_t> &v, size_t limit = 16) {
  std::string s = "{";
  for (size_t i = 0; i < v.size() && i < limit; ++i) s += (i ? ", " : "") + std::to_string(v[i]);
  return s + (v.size() > limit ? ", ...}" : "}");
}
#define EXPECT_TRUE(cond)                                                                                    \
  do {                                                                                                       \
    if (!(cond)) throw std::runtime_error(std::string("expectation failed: ") + #cond + " (line " + std::to_string(__LINE__) + ")"); \
  } while (0)
#define EXPECT_THROWS(stmt)                                                                                  \
  do {                                                                                                       \
    bool threw_ = false;                                                                                     \
    try { stmt; } catch (const std::runtime_error &) { threw_ = true; }                                      \
    if (!threw_) throw std::runtime_error(std::string("expected std::runtime_error from: ") + #stmt);        \
  } while (0)
inline void expectPrefix(const std::vector<int64_t> &got, const std::vector<int64_t> &want) {
  for (size_t i = 0; i < want.size(); ++i)
    if (i >= got.size() || got[i] != want[i])
      throw std::runtime_error("slot " + std::to_string(i) + ": got " + vecToString(got) + " want " + vecToString(want));
}
