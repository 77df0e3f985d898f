// The small routines of the reference's array_router.h / array_static.h that compose from the core operations, on host
// packets (device arrays go through the same templates): hmean, the *_nested and *_inner reductions, any_or / all_or /
// none_or, rad_to_deg / deg_to_rad, abs_dot, copysign_neg / mulsign_neg, fmaddsub / fmsubadd, rol_array / ror_array,
// low / high.
//
//     g++ -O1 -std=c++17 -Iinclude tests/cpp/router_host.cpp -o tests/cpp/router_host.bin
#include <enoki/array.h>
#include <enoki/special.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

using namespace enoki;

#define CHECK(expr) do { if (!(expr)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #expr); exit(1); } } while (0)

int main() {
    using F4 = Array<float, 4>;
    using F3 = Array<float, 3>;
    using N = Array<F4, 3>;                                   // three packets of four lanes
    F4 a(1.f, 2.f, 3.f, 6.f), b(-1.f, 0.5f, 2.f, -2.f), c(10.f, 20.f, 30.f, 40.f);
    N n(a, b, c);

    CHECK(hmean(a) == 3.f && hmean(2.5f) == 2.5f);
    CHECK(hsum_nested(n) == 12.f - 0.5f + 100.f && hprod_nested(F4(1.f, 2.f, 3.f, 4.f)) == 24.f);
    CHECK(hmin_nested(n) == -2.f && hmax_nested(n) == 40.f && hmean_nested(a) == 3.f);
    CHECK(count_nested(n > 1.5f) == 3 + 1 + 4);

    auto si = hsum_inner(n);                                  // one sum per packet
    CHECK(si.coeff(0) == 12.f && si.coeff(1) == -0.5f && si.coeff(2) == 100.f);
    auto mi = hmax_inner(n), ni = hmin_inner(n), pi = hprod_inner(n), ai = hmean_inner(n);
    CHECK(mi.coeff(1) == 2.f && ni.coeff(1) == -2.f && pi.coeff(0) == 36.f && ai.coeff(2) == 25.f);
    CHECK(hsum_inner(a) == 12.f && hsum_inner(7.f) == 7.f);
    auto any_i = any_inner(n > 30.f); auto all_i = all_inner(n > 0.f); auto cnt_i = count_inner(n > 1.5f);
    CHECK(!any_i.coeff(0) && !any_i.coeff(1) && any_i.coeff(2));
    CHECK(all_i.coeff(0) && !all_i.coeff(1) && all_i.coeff(2));
    CHECK(cnt_i.coeff(0) == 3 && cnt_i.coeff(1) == 1 && cnt_i.coeff(2) == 4);
    auto none_i = none_inner(n > 30.f);
    CHECK(none_i.coeff(0) && !none_i.coeff(2));

    CHECK(any_or<false>(a > 5.f) && !any_or<true>(a > 7.f));  // host arrays are evaluated, the default is for device arrays
    CHECK(all_or<false>(a > 0.f) && none_or<false>(a > 7.f) && any_nested_or<false>(n > 39.f) && !all_nested_or<true>(n > 0.f));
    CHECK(none_nested_or<false>(n > 50.f));

    CHECK(std::abs(rad_to_deg(3.14159265f) - 180.f) < 1e-4f && std::abs(deg_to_rad(F3(90.f)).x() - 1.5707964f) < 1e-6f);
    CHECK(abs_dot(F3(1.f, -2.f, 3.f), F3(-1.f, 1.f, -1.f)) == 6.f);
    F4 cs = copysign_neg(a, b), ms = mulsign_neg(a, b);
    CHECK(cs[0] == 1.f && cs[1] == -2.f && cs[3] == 6.f && ms[0] == 1.f && ms[1] == -2.f && ms[2] == -3.f);

    F4 fas = fmaddsub(a, b, c), fsa = fmsubadd(a, b, c);
    CHECK(fas[0] == -1.f - 10.f && fas[1] == 1.f + 20.f && fas[2] == 6.f - 30.f && fas[3] == -12.f + 40.f);
    CHECK(fsa[0] == -1.f + 10.f && fsa[1] == 1.f - 20.f);

    F4 rl = rol_array<1>(a), rr = ror_array<1>(a);
    CHECK(rl[0] == 2.f && rl[3] == 1.f && rr[0] == 6.f && rr[1] == 1.f);
    CHECK(rol_array<4>(a)[2] == a[2] && ror_array<5>(a)[1] == a[0]);

    auto lo = low(a), hi = high(a);
    CHECK(lo.Size == 2 && hi.Size == 2 && lo[1] == 2.f && hi[0] == 3.f);
    auto lo3 = low(F3(1.f, 2.f, 3.f)); auto hi3 = high(F3(1.f, 2.f, 3.f));
    CHECK(lo3.Size == 2 && hi3.Size == 1 && hi3[0] == 3.f);

    // array_math.h: unit_angle / unit_angle_z, prev_float / next_float, isdenormal, polyN
    {
        using V = Array<F4, 3>;                                // four unit vectors at once
        F4 th(0.f, 0.3f, 1.5707964f, 3.0f);
        V u(sin(th), F4(0.f), cos(th)), ez(F4(0.f), F4(0.f), F4(1.f));
        F4 ang = unit_angle(ez, u), angz = unit_angle_z(u);
        for (int i = 0; i < 4; ++i) CHECK(std::abs(ang[i] - th[i]) < 2e-6f && std::abs(angz[i] - th[i]) < 2e-6f);
        F4 x(1.f, -1.f, 0.f, 1e-40f);
        F4 nx = next_float(x), px = prev_float(x);
        CHECK(nx[0] == std::nextafter(1.f, 2.f) && nx[1] == std::nextafter(-1.f, 2.f) && nx[2] == std::nextafter(0.f, 1.f));
        CHECK(px[0] == std::nextafter(1.f, -2.f) && px[1] == std::nextafter(-1.f, -2.f) && px[2] == std::nextafter(0.f, -1.f));
        CHECK(next_float(F4(INFINITY))[0] == INFINITY && std::isnan(prev_float(F4(NAN))[0]));
        auto dn = isdenormal(x);
        CHECK(!dn[0] && !dn[2] && dn[3]);
        F4 t(0.5f);
        CHECK(poly2(t, 1.0, 2.0, 4.0)[0] == 3.f && poly3(t, 1.0, 2.0, 4.0, 8.0)[0] == 4.f && poly4(t, 1.0, 0.0, 0.0, 0.0, 16.0)[0] == 2.f);
    }

    printf("router_host: hmean, nested / inner reductions, *_or, angles, abs_dot, sign transfers, fmaddsub, array rotations, low / high, unit_angle, next / prev_float, polyN\n");
    return 0;
}

// ---- the trait names of array_traits.h that templated user code relies on (compile-time only) ----------------------------
namespace traits_check {
    using F4 = Array<float, 4>;
    using N = Array<F4, 3>;
    static_assert(is_static_array_v<F4> && !is_dynamic_array_v<F4> && array_size_v<F4> == 4 && array_size_v<N> == 3 && array_size_v<float> == 1);
    static_assert(array_depth<N>::value == 2 && is_array<F4>::value && !is_array<float>::value && is_array_any_v<float, F4>);
    static_assert(std::is_same_v<bool_array_t<F4>, Array<bool, 4>> && std::is_same_v<float_array_t<Array<int64_t, 4>>, Array<double, 4>>);
    static_assert(std::is_same_v<size_array_t<F4>, Array<size_t, 4>> && std::is_same_v<array_t<const F4 &>, F4>);
    static_assert(is_std_float_v<double> && is_std_int_v<uint32_t> && is_int64_v<int64_t> && !is_std_type_v<bool> && is_scalar_v<float>);
    static_assert(is_mask<mask_t<F4>>::value && !is_mask<F4>::value && !is_diff_array<F4>::value && !is_cuda_array<F4>::value);
    template <typename T, enable_if_static_array_t<T> = 0> constexpr int pick(const T &) { return 1; }
    template <typename T, enable_if_not_array_t<T> = 0> constexpr int pick(const T &) { return 2; }
    template <typename T, enable_if_std_float_v<T> = 0> constexpr bool fp(T) { return true; }
    static_assert(fp(1.f) && std::is_same_v<identity_t<int>, int>);
}

// ---- log2i / scalar bit counts / sl, sr / scalar_cast / binary_search ------------------------------------------------------
static int run_search_checks() {
    using U4 = Array<uint32_t, 4>;
    CHECK(log2i(1u) == 0 && log2i(255u) == 7 && log2i(256u) == 8 && log2i(uint64_t(1) << 40) == 40);
    CHECK(lzcnt(1u) == 31 && lzcnt(0u) == 32 && tzcnt(8u) == 3 && popcnt(0xF0F0u) == 8 && lzcnt(uint64_t(1)) == 63);
    U4 v(1u, 2u, 255u, 1u << 31);
    U4 l = log2i(v);
    CHECK(l[0] == 0 && l[1] == 1 && l[2] == 7 && l[3] == 31);
    CHECK(sl<3>(v)[1] == 16 && sr<1>(v)[2] == 127 && sl<2>(5u) == 20u);
    CHECK(scalar_cast(Array<float, 1>(2.5f)) == 2.5f && scalar_cast(7) == 7);
    // every lane looks its own value up in a sorted table: first index whose entry is >= the value
    const uint32_t table[10] = { 1, 3, 3, 7, 10, 15, 21, 22, 40, 90 };
    U4 needle(0u, 7u, 23u, 100u);
    U4 found = binary_search(0u, 10u, [&](const U4 &index) {
        Array<bool, 4> m;
        for (size_t i = 0; i < 4; ++i) m[i] = table[std::min<uint32_t>(index[i], 9)] < needle[i];
        return m;
    });
    CHECK(found[0] == 0 && found[1] == 3 && found[2] == 8 && found[3] == 10);
    U4 again = binary_search<U4>(0u, 10u, [&](const auto &index) {        // generic lambda: index type given explicitly
        Array<bool, 4> m;
        for (size_t i = 0; i < 4; ++i) m[i] = table[std::min<uint32_t>(index[i], 9)] < needle[i];
        return m;
    });
    CHECK(again[1] == 3 && again[3] == 10);
    return 0;
}
static const int search_checks_ran = run_search_checks();

// ---- divisor, shape / set_shape / ragged ---------------------------------------------------------------------------------
static int run_shape_checks() {
    using U4 = Array<uint32_t, 4>;
    using I4 = Array<int32_t, 4>;
    U4 v(0u, 13u, 100u, 4000000000u);
    divisor<uint32_t> d7(7u);
    U4 q = v / d7, q2 = d7(v);
    CHECK(q[1] == 1 && q[2] == 14 && q[3] == 4000000000u / 7u && q2[3] == q[3] && (91u / d7) == 13u);
    divisor_ext<int32_t> dm(-3);
    I4 w(-10, -1, 0, 11), r = w % dm, qq = w / dm;
    for (size_t i = 0; i < 4; ++i) CHECK(qq[i] == w[i] / -3 && r[i] == w[i] % -3);
    auto s1 = shape(v);
    auto s2 = shape(Array<U4, 3>(v, v, v));
    CHECK(s1.size() == 1 && s1[0] == 4 && s2.size() == 2 && s2[0] == 3 && s2[1] == 4 && !ragged(v));
    return 0;
}
static const int shape_checks_ran = run_shape_checks();

// ---- stream output ---------------------------------------------------------------------------------------------------------
#include <sstream>
static int run_print_checks() {
    using F3 = Array<float, 3>;
    std::ostringstream a, b, c;
    a << F3(1.f, 2.5f, -3.f);
    CHECK(a.str() == "[1, 2.5, -3]");
    b << Array<F3, 2>(F3(1.f, 2.f, 3.f), F3(4.f, 5.f, 6.f));           // rows = slices of the inner dimension
    CHECK(b.str() == "[[1, 4],\n [2, 5],\n [3, 6]]");
    c << (F3(1.f, 2.f, 3.f) > 1.5f);
    CHECK(c.str() == "[0, 1, 1]");
    return 0;
}
static const int print_checks_ran = run_print_checks();

// ---- transform: conflict-free read-modify-write -------------------------------------------------------------------------
static int run_transform_checks() {
    using F4 = Array<float, 4>;
    using U4 = Array<uint32_t, 4>;
    float hist[4] = { 0.f, 0.f, 0.f, 0.f };
    U4 bin(1u, 3u, 1u, 1u);                                  // three lanes hit the same bin
    transform<F4>(hist, bin, [](float &slot, float w, bool) { slot += w; }, F4(1.f, 2.f, 4.f, 8.f));
    CHECK(hist[1] == 13.f && hist[3] == 2.f && hist[0] == 0.f);
    transform<F4>(hist, bin, [](float &slot, float w, bool) { slot *= w; }, F4(2.f), Array<bool, 4>(true, false, true, false));
    CHECK(hist[1] == 52.f && hist[3] == 2.f);
    transform<float>(hist, 2u, [](float &slot, bool) { slot = 7.f; });
    CHECK(hist[2] == 7.f);
    return 0;
}
static const int transform_checks_ran = run_transform_checks();

static int run_slice_checks() {
    using F3 = Array<float, 3>;
    F3 v(1.f, 2.f, 3.f);
    auto s = slice(Array<F3, 2>(v, v * 2.f), 1);            // static arrays have no dynamic dimension: returned unchanged in shape
    CHECK(s.coeff(1).coeff(2) == 6.f && slice(5.f, 9) == 5.f);
    int unused = 0; ENOKI_MARK_USED(unused);
    return 0;
}
static const int slice_checks_ran = run_slice_checks();
