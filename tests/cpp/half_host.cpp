// enoki/half.h and the load / store functions of enoki/array.h on the host.
//
//  * all 65536 binary16 encodings: half -> float is exact (checked against the definition evaluated in double) and
//    float -> half returns the same bits (NaNs: stay NaN, keep the sign, become quiet);
//  * float -> half rounding: 2^23 floats spread over the whole encoding space plus every boundary case (ties, the
//    subnormal range, overflow) against a reference rounding done in double arithmetic -- and against the F16C
//    instruction when the build machine has it (what the reference's x86 build uses, half.h:112-114);
//  * the reference's own check (tests/float.cpp:215-238): Array<half, 4> <-> Array<float, 4> through load / store agrees
//    with the scalar conversions for every encoding;
//  * load / store / masked forms on scalars, packets and nested arrays.
//
//     g++ -O1 -std=c++17 -Iinclude tests/cpp/half_host.cpp -o tests/cpp/half_host.bin     (adds -mf16c when available)
#include <enoki/array.h>
#include <enoki/half.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#if defined(__F16C__)
#  include <immintrin.h>
#endif

using namespace enoki;

#define CHECK(expr) do { if (!(expr)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #expr); exit(1); } } while (0)

static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/// value of an encoding, from the definition
static double half_value(uint16_t h) {
    const int e = (h >> 10) & 31, m = h & 1023;
    const double s = (h & 0x8000) ? -1.0 : 1.0;
    if (e == 0) return s * std::ldexp((double) m, -24);
    if (e == 31) return m ? NAN : s * INFINITY;
    return s * std::ldexp(1.0 + m / 1024.0, e - 15);
}

/// round to nearest even by search over the two neighbouring encodings (independent of the code under test)
static uint16_t reference_round(float f) {
    if (std::isnan(f)) return 0x7E00;
    const uint16_t sign = std::signbit(f) ? 0x8000 : 0;
    const double a = std::fabs((double) f);
    if (a >= 65520.0) return sign | 0x7C00;
    uint16_t lo = 0, hi = 0x7C00;                 // largest encoding <= a by bisection (encodings are monotonic)
    while (hi - lo > 1) {
        uint16_t mid = (uint16_t) ((lo + hi) / 2);
        if (half_value(mid) <= a) lo = mid; else hi = mid;
    }
    const double dl = a - half_value(lo), dh = (hi == 0x7C00 ? 65536.0 : half_value(hi)) - a;
    uint16_t r = dl < dh ? lo : dh < dl ? hi : ((lo & 1) ? hi : lo);
    return sign | r;
}

int main() {
    // ---- every encoding ---------------------------------------------------------------------------------------------
    for (uint32_t i = 0; i < 0x10000; ++i) {
        const uint16_t h = (uint16_t) i;
        const float f = (float) half::from_binary(h);
        const double v = half_value(h);
        if (std::isnan(v)) {
            CHECK(std::isnan(f) && std::signbit(f) == ((h & 0x8000) != 0) && (bits(f) & 0x00400000u));
            const uint16_t back = half(f).value;
            CHECK((back & 0x7C00) == 0x7C00 && (back & 0x03FF) != 0 && (back & 0x8000) == (h & 0x8000));
        } else {
            CHECK((double) f == v && std::signbit(f) == std::signbit(v));
            CHECK(half(f).value == h);
        }
#if defined(__F16C__)
        const float hw = _mm_cvtss_f32(_mm_cvtph_ps(_mm_cvtsi32_si128((int) h)));
        CHECK(bits(hw) == bits(f));
#endif
    }
    // ---- rounding ---------------------------------------------------------------------------------------------------
    size_t checked = 0;
    auto check_round = [&](float f) {
        const uint16_t got = half::float32_to_float16(f);
        if (std::isnan(f)) { CHECK((got & 0x7C00) == 0x7C00 && (got & 0x3FF)); return; }
        CHECK(got == reference_round(f));
#if defined(__F16C__)
        const uint16_t hw = (uint16_t) _mm_cvtsi128_si32(_mm_cvtps_ph(_mm_set_ss(f), _MM_FROUND_TO_NEAREST_INT));
        CHECK(hw == got);
#endif
        ++checked;
    };
    for (uint64_t u = 0; u < (1ull << 32); u += 509) check_round(from_bits((uint32_t) u));          // 8.4 M floats, odd stride
    for (uint32_t h = 0; h < 0x7C00; ++h) {                                                          // around every tie
        const double lo = half_value((uint16_t) h), hi = h + 1 == 0x7C00 ? 65536.0 : half_value((uint16_t) (h + 1));
        const float mid = (float) (0.5 * (lo + hi));
        for (int d = -2; d <= 2; ++d) {
            const float f = from_bits(bits(mid) + (uint32_t) d);
            check_round(f); check_round(-f);
        }
    }
#if defined(__F16C__)
    for (uint64_t u = 0; u < (1ull << 32); u += 61) {                 // 70 M floats against the instruction alone (fast)
        const float f = from_bits((uint32_t) u);
        const uint16_t hw = (uint16_t) _mm_cvtsi128_si32(_mm_cvtps_ph(_mm_set_ss(f), _MM_FROUND_TO_NEAREST_INT));
        const uint16_t got = half::float32_to_float16(f);
        if (std::isnan(f)) CHECK((got & 0x7FFF) > 0x7C00 && (hw & 0x7FFF) > 0x7C00 && got == hw); else CHECK(got == hw);
    }
#endif
    // ---- arithmetic goes through float --------------------------------------------------------------------------------
    CHECK((float) (half(1.5f) + half(2.25f)) == 3.75f && (float) (half(3.f) * half(0.5f)) == 1.5f);
    CHECK((float) (-half(2.f)) == -2.f && (float) (1.f / half(4.f)) == 0.25f && half(1.f) < half(2.f) && half(2.f) == half(2.f));
    CHECK((float) (half(2049.f)) == 2048.f && (float) half(2051.f) == 2052.f);                      // ties to even
    CHECK(std::numeric_limits<half>::max().value == 0x7BFF && (float) std::numeric_limits<half>::epsilon() == 0x1p-10f);
    // ---- the reference's test (tests/float.cpp:215-238) ---------------------------------------------------------------
    using T = Array<float, 4>;
    using THalf = Array<half, 4>;
    for (uint32_t i = 0; i < 0xFFFF; ++i) {
        uint16_t data[8] = { (uint16_t) i }, data3[8] = { (uint16_t) i };
        float f1 = T(load<THalf>((const half *) data))[0];
        float f2 = (float) half::from_binary(data[0]);
        bool both_nan = std::isnan(f1) && std::isnan(f2);
        CHECK(bits(f1) == bits(f2) || both_nan);
        half data2[8];
        store(data2, THalf(T(f1)));
        data3[0] = half(f2).value;
        CHECK(data2[0].value == data3[0] || both_nan);
    }
    // ---- load / store ---------------------------------------------------------------------------------------------------
    {
        float mem[12];
        for (int i = 0; i < 12; ++i) mem[i] = (float) i + 0.5f;
        auto p = load<Array<float, 4>>(mem + 1);
        CHECK(p[0] == 1.5f && p[3] == 4.5f);
        Array<bool, 4> m(true, false, true, false);
        auto q = load_unaligned<Array<float, 4>>(mem, m);
        CHECK(q[0] == 0.5f && q[1] == 0.f && q[2] == 2.5f && q[3] == 0.f);
        auto n = load<Array<Array<float, 4>, 3>>(mem);                 // component after component
        CHECK(n[0][0] == 0.5f && n[1][0] == 4.5f && n[2][3] == 11.5f);
        float out[12] = { };
        store(out, n);
        for (int i = 0; i < 12; ++i) CHECK(out[i] == mem[i]);
        float out2[4] = { 9.f, 9.f, 9.f, 9.f };
        store_unaligned(out2, p, m);
        CHECK(out2[0] == 1.5f && out2[1] == 9.f && out2[2] == 3.5f && out2[3] == 9.f);
        CHECK(load<float>(mem + 2) == 2.5f);
        float one = 0.f; store(&one, 7.f); CHECK(one == 7.f);
    }
    printf("half_host: 65536 encodings, %zu roundings against an independent reference%s, Array<half, 4> <-> Array<float, 4>, "
           "load / store\n", checked,
#if defined(__F16C__)
           " and against F16C (70 M more)"
#else
           ""
#endif
    );
    return 0;
}
