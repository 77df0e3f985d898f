// The reference's OWN headers (array.h, array_router.h, array_math.h, array_struct.h, dynamic.h from /root/reference/include)
// driving this repository's device backend through integration/enoki/hip.h -- the header a maintainer of the reference
// would add next to cuda.h -- and, in the same binary, the reference's CPU arrays on the same inputs.  Every templated
// function below is instantiated twice: on DynamicArray<Packet<float>> (the reference's CPU path: the parity target) and
// on HIPArray<float> (the reference's router -> member concept -> C ABI -> HIP kernels); results are compared bit for bit
// where the operation is class A, to the documented bounds otherwise.
//
//   g++ -std=c++17 -O2 -mavx2 -mfma ... -ffp-contract=off -I/root/reference/include -Iintegration -Iinclude \
//       tests/cpp/reference_side_hip.cpp integration/hip_hooks.cpp -Lenoki_amd -lenoki-hip
// Built only where /root/reference exists (enoki_amd/_build.py); the binary travels to the GPU box.
#include <enoki/array.h>
#include <enoki/dynamic.h>
#include <enoki/hip.h>
#include <enoki/special.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

using namespace enoki;

using FloatP = Packet<float>;
using FloatX = DynamicArray<FloatP>;
using UInt32X = DynamicArray<Packet<uint32_t>>;
using FloatH = HIPArray<float>;
using UInt32H = HIPArray<uint32_t>;

static int g_failures = 0, g_checks = 0;

template <typename T> static std::vector<T> host(const DynamicArray<Packet<T>> &a) {
    std::vector<T> v(a.size());
    for (size_t i = 0; i < v.size(); ++i) v[i] = a.coeff(i);
    return v;
}
template <typename T> static std::vector<T> host(const HIPArray<T> &a) {
    std::vector<T> v(a.size());
    if (!v.empty()) ek_hip_memcpy_to_host(v.data(), a.data(), v.size() * sizeof(T));
    return v;
}

template <typename T> static void expect_bits(const char *what, const std::vector<T> &cpu, const std::vector<T> &dev) {
    ++g_checks;
    bool ok = cpu.size() == dev.size() && (cpu.empty() || memcmp(cpu.data(), dev.data(), cpu.size() * sizeof(T)) == 0);
    if (!ok) {
        ++g_failures;
        size_t bad = 0, first = (size_t) -1;
        for (size_t i = 0; i < std::min(cpu.size(), dev.size()); ++i)
            if (memcmp(&cpu[i], &dev[i], sizeof(T)) != 0) { if (first == (size_t) -1) first = i; ++bad; }
        printf("  MISMATCH %s: sizes %zu / %zu, %zu entries differ (first at %zu)\n", what, cpu.size(), dev.size(), bad, first);
    } else {
        printf("  ok  %-44s bit-identical (%zu entries)\n", what, cpu.size());
    }
}
static void expect_close(const char *what, const std::vector<float> &cpu, const std::vector<float> &dev, double rel) {
    ++g_checks;
    double worst = 0;
    bool ok = cpu.size() == dev.size();
    for (size_t i = 0; ok && i < cpu.size(); ++i) {
        double e = std::fabs((double) cpu[i] - dev[i]) / std::max(1e-30, std::fabs((double) cpu[i]));
        worst = std::max(worst, e);
    }
    ok = ok && worst <= rel;
    if (!ok) { ++g_failures; printf("  MISMATCH %s: max relative difference %.3g > %.3g\n", what, worst, rel); }
    else printf("  ok  %-44s max relative difference %.2g (bound %.1g)\n", what, worst, rel);
}

// ---- the "user code": templated on the array type, routed by the reference's headers ----------------------------------
template <typename Float> Float arithmetic(const Float &x, const Float &y) {
    return fmadd(x, y, 0.5f) * (x - y) / (abs(y) + 1.f) + sqrt(abs(x)) - min(x, y) * max(x, 0.25f);
}
template <typename Float> Float transcendental(const Float &x) { return sin(x) * exp(x * 0.25f) + cos(x) - log(abs(x) + 1.f); }
// the second wave: the reference composes these from primitives with the array type's own operations (array_math.h) -- on
// HIPArray that is dozens of kernels per function, all dispatched by the reference's code
template <typename Float> Float second_wave_exact(const Float &x, const Float &y) {
    Float u = x * 0.3f;                                     // |u| < 1
    return asin(u) + acos(u) * atan(x) - atan2(y, x) + asinh(x) * cbrt(y) + pow(abs(x) + 0.5f, y) + erf(u) + erfinv(u);
}
template <typename Float> Float second_wave_rcp(const Float &x) {
    Float u = x * 0.4f;                                     // away from the poles of tan
    return tan(u) + sinh(x) * 0.1f + cosh(x) * 0.1f + tanh(x) + 4.f;
}
template <typename Float> Float branches(const Float &x, const Float &y) {
    auto m = (x > y) & (x * y < 0.5f);
    Float r = select(m, x * 2.f, y - 1.f);
    r[r < -1.f] = -1.f;                                     // masked assignment (array_base.h / array_masked.h)
    return r + floor(x) - ceil(y) + round(x * y);
}
template <typename Float, typename UInt32 = uint32_array_t<Float>> UInt32 integers(const Float &x) {
    UInt32 i = UInt32(abs(x) * 1000.f);
    return ((i << 3) ^ (i >> 1)) + popcnt(i) * 7u + (i & 0xffu) * (i | 1u);
}
template <typename Float, typename UInt32 = uint32_array_t<Float>> Float indexed(const Float &table, const UInt32 &idx, const Float &x) {
    return gather<Float>(table, idx, x > 0.f) * x;          // array_struct.h wrappers (set_scatter_gather_operand hooks)
}
template <typename Float, typename UInt32 = uint32_array_t<Float>> Float scattered(const Float &x, size_t n) {
    // a permutation (unique targets: duplicates would leave the winner unspecified on a device), then an integer-valued
    // scatter_add through colliding indices (exact in float32 whatever the order)
    UInt32 perm = (arange<UInt32>(n) * 7919u) % UInt32((uint32_t) n);       // 7919 is coprime to n = 100003
    Float t = zero<Float>(n);
    scatter(t, x, perm, x < 2.f);
    Float counts = zero<Float>(1024);
    scatter_add(counts, floor(abs(x) * 3.f), perm & 1023u);
    return t + gather<Float>(counts, perm & 1023u);
}
template <typename Float> struct Sample { Float a, b; };

template <typename Float, typename UInt32> static void fill(size_t n, size_t K, Float &x, Float &y, Float &table, UInt32 &idx) {
    x = linspace<Float>(-3.f, 3.f, n);
    y = sin(linspace<Float>(0.f, 40.f, n)) * 1.5f;
    table = cos(linspace<Float>(0.f, 10.f, K));
    idx = (arange<UInt32>(n) * 2654435761u) >> 20;           // < 4096
}

int main() {
    if (ek_hip_init(-1) != EK_OK) { printf("ek_hip_init failed: %s\n", ek_hip_last_error()); return 2; }
    static_assert(is_cuda_array_v<FloatH> && is_dynamic_array_v<FloatH> && !is_cuda_array_v<FloatX>);
    static_assert(std::is_same_v<mask_t<FloatH>, HIPArray<bool>> && std::is_same_v<uint32_array_t<FloatH>, UInt32H>);
    const size_t n = 100003, K = 4096;
    FloatX xc, yc, tc; UInt32X ic;
    FloatH xd, yd, td; UInt32H id;
    fill(n, K, xc, yc, tc, ic);
    fill(n, K, xd, yd, td, id);
    printf("reference headers + integration/enoki/hip.h, %zu elements\n", n);
    expect_bits("inputs: linspace / arange / integer hash", host(ic), host(id));
    // inputs contain sin / cos: use the CPU values on both sides from here on so that every check isolates ONE function
    xd = FloatH::copy(xc.data(), n); yd = FloatH::copy(yc.data(), n); td = FloatH::copy(tc.data(), K);

    expect_bits("arithmetic (fmadd, div, sqrt, min, max)", host(arithmetic(xc, yc)), host(arithmetic(xd, yd)));
    expect_bits("transcendental (sin, cos, exp, log)", host(transcendental(xc)), host(transcendental(xd)));
    expect_bits("asin acos atan atan2 asinh cbrt pow erf erfinv", host(second_wave_exact(xc, yc)), host(second_wave_exact(xd, yd)));
    expect_close("tan sinh cosh tanh (contain rcp: class C)", host(second_wave_rcp(xc)), host(second_wave_rcp(xd)), 2e-5);
    expect_bits("compare / select / masked assign / rounding", host(branches(xc, yc)), host(branches(xd, yd)));
    expect_bits("integer ops (shifts, popcnt, and / or / xor, mul)", host(integers(xc)), host(integers(xd)));
    expect_bits("masked gather through array_struct.h", host(indexed(tc, ic, xc)), host(indexed(td, id, xd)));
    expect_bits("masked scatter (permutation) + scatter_add", host(scattered(xc, n)), host(scattered(xd, n)));
    {
        // horizontal operations: order-dependent in floating point (class D), exact on integers
        auto hc = hsum(arithmetic(xc, yc)); auto hd = hsum(arithmetic(xd, yd));
        expect_close("hsum (float32, 100003 terms)", std::vector<float>{ hc }, host(hd), 2e-4);
        expect_bits("hmax / hmin", std::vector<float>{ hmax(xc * yc), hmin(xc * yc) }, std::vector<float>{ host(hmax(xd * yd))[0], host(hmin(xd * yd))[0] });
        expect_bits("hsum of integers", std::vector<uint32_t>{ hsum(integers(xc)) }, host(hsum(integers(xd))));
        ++g_checks;
        size_t cc = count(xc > yc), cd = count(xd > yd);
        bool ac = any(xc > 2.9f), ad = any(xd > 2.9f), lc = all(xc > -4.f), ld = all(xd > -4.f);
        if (cc != cd || ac != ad || lc != ld) { ++g_failures; printf("  MISMATCH count / any / all\n"); }
        else printf("  ok  %-44s %zu, %d, %d\n", "count / any / all", cd, (int) ad, (int) ld);
    }
    {
        // nested arrays over the backend: Array<FloatH, 3> through the reference's static-array machinery
        using Vector3c = Array<FloatX, 3>; using Vector3d = Array<FloatH, 3>;
        Vector3c vc(xc, yc, xc * yc); Vector3d vd(xd, yd, xd * yd);
        expect_bits("Array<HIPArray, 3>: dot . cross, squared_norm",
                    host(FloatX(dot(vc + 1.f, cross(vc, Vector3c(1.f, 2.f, 3.f))) + squared_norm(vc))),
                    host(FloatH(dot(vd + 1.f, cross(vd, Vector3d(1.f, 2.f, 3.f))) + squared_norm(vd))));
        // normalize() goes through rsqrt: rsqrtps + one Newton step on the CPU, an exact 1 / sqrt on the device (class C)
        expect_close("Array<HIPArray, 3>: normalize (rsqrt, class C)", host(FloatX(normalize(vc + 2.f).y())), host(FloatH(normalize(vd + 2.f).y())), 1e-6);
    }
    {
        // float64 through the same code
        using DoubleX = DynamicArray<Packet<double>>; using DoubleH = HIPArray<double>;
        DoubleX ac = linspace<DoubleX>(-3.0, 3.0, 50021), bc = cos(ac * 7.0);
        DoubleH ad = DoubleH::copy(ac.data(), ac.size()), bd = DoubleH::copy(bc.data(), bc.size());
        auto f64 = [](const auto &a, const auto &b) { return fmadd(sin(a), exp(b), log(abs(a) + 1.0)) / (sqrt(abs(b)) + 1.0) + atan2(a, b) + tanh(a); };
        std::vector<double> c1(ac.size()), d1(ac.size());
        auto rc = f64(ac, bc); auto rd = f64(ad, bd);
        for (size_t i = 0; i < c1.size(); ++i) c1[i] = rc.coeff(i);
        ek_hip_memcpy_to_host(d1.data(), rd.data(), d1.size() * sizeof(double));
        expect_bits("float64: sin exp log sqrt div atan2 tanh", c1, d1);
    }
    printf("%d/%d checks passed\n", g_checks - g_failures, g_checks);
    return g_failures ? 1 : 0;
}
