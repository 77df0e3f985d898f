// tests/cpp/refshim/test.h -- the few lines of test harness that the reference's tests/*.cpp expect from their
// "test.h", written for ONE purpose: compile the reference's OWN test sources (e.g. /root/reference/tests/autodiff.cpp,
// unmodified, #included where it lies) against THIS repository's headers with the device array types substituted for the
// CPU ones (refshim/enoki/dynamic.h maps DynamicArray<Packet<T>> to HIPArray<T>).  A failing assert() throws instead
// of aborting so that one run reports every test: "N/M passed" on the last line, failures listed with their cause.
#pragma once

#include <enoki/array.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

using namespace enoki;

namespace test {
struct Failure : std::runtime_error { using std::runtime_error::runtime_error; };

using TestFn = void (*)();
inline std::vector<std::pair<std::string, TestFn>> &registry() {
    static std::vector<std::pair<std::string, TestFn>> r;
    return r;
}
struct Test {
    Test(const char *name, TestFn fn) { registry().emplace_back(name, fn); }
};

/// assert() accepts what the reference's tests hand it on EITHER backend: bool on the CPU arrays (hmax() of a
/// DynamicArray is a scalar), a size-1 / size-n mask array on a device backend (hmax() is a size-1 array there)
inline bool truth(bool v) { return v; }
template <typename T, enoki::enable_if_t<enoki::is_array_v<T>> = 0> bool truth(const T &m) { return enoki::all(m); }

inline int run_all() {
    auto tests = registry();
    std::sort(tests.begin(), tests.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    size_t passed = 0;
    std::vector<std::string> failures;
    for (auto &t : tests) {
        std::cout << "    " << t.first << ".. " << std::flush;
        std::ostringstream sink;                      // the tests print graphviz dumps: keep the log readable
        std::streambuf *saved = std::cout.rdbuf(sink.rdbuf());
        std::string error;
        try {
            t.second();
        } catch (const std::exception &e) {
            error = e.what();
        }
        std::cout.rdbuf(saved);
        if (error.empty()) { std::cout << "ok." << std::endl; ++passed; }
        else { std::cout << "FAILED: " << error << std::endl; failures.push_back(t.first + ": " + error); }
    }
    std::cout << std::endl;
    for (const std::string &f : failures) std::cout << "  failure -- " << f << std::endl;
    std::cout << passed << "/" << tests.size() << " passed" << std::endl;
    return failures.empty() ? 0 : 1;
}
} // namespace test

#define ENOKI_TEST(name) void name(); static test::Test name##_test{#name, &name}; void name()

#include <test_assert.h>
