// Tiny gtest-shaped assertion macros (gtest is not in the image) so that the
// host tests read like the reference's *_unittest.cpp files.
#ifndef GLX_HOST_TEST_UTIL_H_
#define GLX_HOST_TEST_UTIL_H_
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

struct TestCase {
  std::string name;
  std::function<void()> body;
};
inline std::vector<TestCase>& AllTests() {
  static std::vector<TestCase> t;
  return t;
}
inline int& Failures() {
  static int f = 0;
  return f;
}
struct TestRegistrar {
  TestRegistrar(const std::string& n, std::function<void()> b) { AllTests().push_back({n, b}); }
};
#define TEST(Suite, Name)                                                         \
  static void Suite##_##Name();                                                   \
  static TestRegistrar reg_##Suite##_##Name(#Suite "." #Name, Suite##_##Name);    \
  static void Suite##_##Name()

#define EXPECT_TRUE(c)                                                            \
  do {                                                                            \
    if (!(c)) {                                                                   \
      std::printf("  FAILED %s:%d: %s\n", __FILE__, __LINE__, #c);                \
      ++Failures();                                                               \
    }                                                                             \
  } while (0)
#define EXPECT_EQ(a, b) EXPECT_TRUE((a) == (b))
#define EXPECT_FLOAT_EQ(a, b) EXPECT_TRUE(std::fabs((double)(a) - (double)(b)) <= 4e-7 * std::fabs((double)(b)))

inline int RunAllTests() {
  std::setvbuf(stdout, nullptr, _IOLBF, 0);  // a run that dies (or is killed by a tool) still shows how far it got
  for (auto& t : AllTests()) {
    int before = Failures();
    std::printf("[ RUN  ] %s\n", t.name.c_str());
    t.body();
    std::printf("[ %s ] %s\n", Failures() == before ? " OK " : "FAIL", t.name.c_str());
  }
  std::printf("%d test(s), %d failure(s)\n", (int)AllTests().size(), Failures());
  std::fflush(stdout);
  return Failures() == 0 ? 0 : 1;
}
#endif
