// Fused == unfused, bit for bit: every floating point function a vectorize() kernel can call is evaluated once inside a
// fused kernel (one-element packets, include/enoki/vectorize.h) and once op by op on HIPArray; the results are compared
// on the device.  hipcc translation unit (built by enoki_amd/_build.py into tests/cpp/libvectorize_math_hip.so), driven by
// tests/test_sphere_gpu.py::test_fused_math_matches_kernels.
#include <enoki/vectorize.h>
ENOKI_DEVICE_CODE_BEGIN
#include <enoki/special.h>
#include <enoki/stl.h>
ENOKI_DEVICE_CODE_END

#include <cstdio>

using namespace enoki;

ENOKI_DEVICE_CODE_BEGIN
// one functor per function so that the same source text instantiates on packets (fused) and on device arrays (eager)
#define MATH_CASE(name, expr)                                                                          \
    struct case_##name { template <typename T> auto operator()(const T &x, const T &y) const { (void) y; return expr; } };
MATH_CASE(sin, sin(x))       MATH_CASE(cos, cos(x))       MATH_CASE(tan, tan(x))       MATH_CASE(exp, exp(x))
MATH_CASE(log, log(abs(x)))  MATH_CASE(asin, asin(x * 0.25f)) MATH_CASE(acos, acos(x * 0.25f)) MATH_CASE(atan, atan(x))
MATH_CASE(sinh, sinh(x))     MATH_CASE(cosh, cosh(x))     MATH_CASE(tanh, tanh(x))     MATH_CASE(cbrt, cbrt(x))
MATH_CASE(atan2, atan2(x, y)) MATH_CASE(pow, pow(abs(x), y)) MATH_CASE(sqrt, sqrt(abs(x))) MATH_CASE(rsqrt, rsqrt(abs(x)))
MATH_CASE(rcp, rcp(x))       MATH_CASE(div, x / y)        MATH_CASE(fma, fmadd(x, y, x))
MATH_CASE(mix, sin(x) * exp(y * 0.5f) + log(abs(x) + 1.f) / (cos(y) + 2.f))
// elliptic integrals: inside the fused kernel the duplication loop runs per lane (include/enoki/ellint.h)
MATH_CASE(ellint_1, ellint_1(x, y * 0.25f))   MATH_CASE(ellint_2, ellint_2(x, y * 0.25f))
MATH_CASE(ellint_3, ellint_3(x, y * 0.25f, abs(x) * 0.2f))   MATH_CASE(comp_ellint_1, comp_ellint_1(y * 0.25f))
MATH_CASE(carlson_rd, carlson_rd(x * x, abs(y) + 0.5f, abs(x) + 1.f))
struct case_sincos { template <typename T> auto operator()(const T &x, const T &y) const { auto [s, c] = sincos(x); return s * y + c; } };
ENOKI_DEVICE_CODE_END

template <typename Float, typename Case> static size_t mismatches(const Float &x, const Float &y, const Case &f) {
    Float fused = vectorize([f](auto &&a, auto &&b) { return f(a, b); }, x, y);
    Float eager = f(x, y);
    using UInt = HIPArray<std::conditional_t<sizeof(scalar_t<Float>) == 4, uint32_t, uint64_t>>;
    const size_t bad = count(neq(reinterpret_array<UInt>(fused), reinterpret_array<UInt>(eager)));
    if (bad && getenv("VECTORIZE_MATH_VERBOSE"))
        fprintf(stderr, "  x=%.9g y=%.9g fused=%.17g eager=%.17g (sizes %zu %zu)\n", (double) x.coeff(0), (double) y.coeff(0),
                (double) fused.coeff(0), (double) eager.coeff(0), fused.size(), eager.size());
    return bad;
}

/// std::pair / std::tuple / std::array as results and as arguments of a fused kernel (include/enoki/stl.h)
template <typename Float> static size_t stl_mismatches(const Float &x, const Float &y) {
    using UInt = HIPArray<std::conditional_t<sizeof(scalar_t<Float>) == 4, uint32_t, uint64_t>>;
    auto differ = [](const Float &a, const Float &b) { return count(neq(reinterpret_array<UInt>(a), reinterpret_array<UInt>(b))); };
    std::pair<Float, Float> sc = vectorize([](auto &&a) { return sincos(a); }, x);
    auto ref = sincos(x);
    size_t bad = differ(sc.first, ref.first) + differ(sc.second, ref.second);
    std::tuple<Float, Float, Float> t = vectorize([](auto &&a, auto &&b) { return std::make_tuple(a + b, a * b, fmadd(a, b, a)); }, x, y);
    bad += differ(std::get<0>(t), x + y) + differ(std::get<1>(t), x * y) + differ(std::get<2>(t), fmadd(x, y, x));
    Float packed = vectorize([](auto &&p) { return p.first - p.second; }, sc);             // a pair as a sliced ARGUMENT
    bad += differ(packed, ref.first - ref.second);
    std::array<Float, 3> arr = vectorize([](auto &&a, auto &&b) {                          // std::array as a result ...
        using P = std::decay_t<decltype(a)>;
        return std::array<P, 3>{ { a + b, a - b, a * b } };
    }, x, y);
    bad += differ(arr[0], x + y) + differ(arr[1], x - y) + differ(arr[2], x * y);
    Float folded = vectorize([](auto &&v) { return fmadd(v[0], v[1], v[2]); }, arr);       // ... and as a sliced argument
    return bad + differ(folded, fmadd(x + y, x - y, x * y));
}

template <typename Scalar> static int run(size_t n, char *report, size_t report_size) {
    using Float = HIPArray<Scalar>;
    // (not linspace: one slice would divide by n - 1 = 0)
    Float t = arange<Float>(n) * Float(Scalar(1) / Scalar(n));
    Float x = fmadd(t, Float(Scalar(8)), Float(Scalar(-4))), y = fmadd(t, Float(Scalar(-5.75)), Float(Scalar(3.5)));
    int bad = 0;
    size_t used = 0;
#define RUN_CASE(name) { size_t m = mismatches(x, y, case_##name()); if (m) ++bad;                                  \
        used += (size_t) snprintf(report + used, used < report_size ? report_size - used : 0, "%s:%zu ", #name, m); }
    RUN_CASE(sin) RUN_CASE(cos) RUN_CASE(tan) RUN_CASE(exp) RUN_CASE(log) RUN_CASE(asin) RUN_CASE(acos) RUN_CASE(atan)
    RUN_CASE(sinh) RUN_CASE(cosh) RUN_CASE(tanh) RUN_CASE(cbrt) RUN_CASE(atan2) RUN_CASE(pow) RUN_CASE(sqrt) RUN_CASE(rsqrt)
    RUN_CASE(rcp) RUN_CASE(div) RUN_CASE(fma) RUN_CASE(mix) RUN_CASE(sincos)
    RUN_CASE(ellint_1) RUN_CASE(ellint_2) RUN_CASE(ellint_3) RUN_CASE(comp_ellint_1) RUN_CASE(carlson_rd)
    { size_t m = stl_mismatches(x, y); if (m) ++bad;
      used += (size_t) snprintf(report + used, used < report_size ? report_size - used : 0, "pair_tuple:%zu ", m); }
    return bad;
}

extern "C" int vectorize_math_check(int is_double, size_t n, char *report, size_t report_size) {
    try {
        return is_double ? run<double>(n, report, report_size) : run<float>(n, report, report_size);
    } catch (const std::exception &e) {
        snprintf(report, report_size, "exception: %s", e.what());
        return -1;
    }
}
