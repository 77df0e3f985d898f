/*
    enoki/half.h -- IEEE 754 binary16 storage type (reference: include/enoki/half.h)

    `enoki::half` holds 16 bits and computes through float: every operator converts, works in single precision and
    rounds back (round to nearest even, what the reference's F16C build does, half.h:112-114).  The conversions are
    plain integer code that runs on the host and inside device kernels alike; NaNs are quieted and keep their upper
    payload bits, values of magnitude >= 65520 become infinity, results below 2^-24 round to +-0 or the smallest
    subnormal.  tests/cpp/half_host.cpp checks all 65536 encodings, and the rounding of 8.7 M floats against an
    independent reference rounding and of 70 M more against the hardware conversion (F16C) where the build machine has it.

    Static arrays of halves (Array<half, N>) convert to and from arrays of floats component by component.
*/
#pragma once

#include <cstdint>
#include <cstring>
#include <limits>
#include <ostream>
#include <type_traits>

namespace enoki { struct half; }

namespace std {
    template <> struct is_floating_point<enoki::half> : true_type { };
    template <> struct is_arithmetic<enoki::half> : true_type { };
    template <> struct is_signed<enoki::half> : true_type { };
}

namespace enoki {

struct half {
    uint16_t value;

    half() : value(0x7FFF) { }                       // a NaN, so that uninitialised use shows (half.h:32-36, debug builds)

    template <typename T, std::enable_if_t<std::is_arithmetic_v<T> && !std::is_same_v<T, half>, int> = 0>
    half(T v) : value(float32_to_float16((float) v)) { }

    template <typename T, std::enable_if_t<std::is_arithmetic_v<T> && !std::is_same_v<T, half>, int> = 0>
    operator T() const { return (T) float16_to_float32(value); }

    static half from_binary(uint16_t bits) { half h; h.value = bits; return h; }

    half operator+(half h) const { return half((float) *this + (float) h); }
    half operator-(half h) const { return half((float) *this - (float) h); }
    half operator*(half h) const { return half((float) *this * (float) h); }
    half operator/(half h) const { return half((float) *this / (float) h); }
    half operator-() const { return from_binary(value ^ 0x8000); }

#define ENOKI_HALF_MIXED(op)                                                                             \
    template <typename T, std::enable_if_t<std::is_arithmetic_v<T> && !std::is_same_v<T, half>, int> = 0> \
    friend half operator op(T a, half b) { return half(a) op b; }
    ENOKI_HALF_MIXED(+) ENOKI_HALF_MIXED(-) ENOKI_HALF_MIXED(*) ENOKI_HALF_MIXED(/)
#undef ENOKI_HALF_MIXED

    half &operator+=(half h) { return *this = *this + h; }
    half &operator-=(half h) { return *this = *this - h; }
    half &operator*=(half h) { return *this = *this * h; }
    half &operator/=(half h) { return *this = *this / h; }

    bool operator==(half h) const { return (float) *this == (float) h; }
    bool operator!=(half h) const { return (float) *this != (float) h; }
    bool operator<(half h) const { return (float) *this < (float) h; }
    bool operator>(half h) const { return (float) *this > (float) h; }
    bool operator<=(half h) const { return (float) *this <= (float) h; }
    bool operator>=(half h) const { return (float) *this >= (float) h; }

    friend std::ostream &operator<<(std::ostream &os, const half &h) { return os << (float) h; }

    /// float -> binary16, round to nearest even
    static uint16_t float32_to_float16(float f) {
        uint32_t x;
        memcpy(&x, &f, 4);
        const uint16_t sign = (uint16_t) ((x >> 16) & 0x8000u);
        x &= 0x7FFFFFFFu;
        if (x >= 0x7F800000u)                                       // inf / NaN (quieted, upper payload bits kept)
            return (uint16_t) (sign | 0x7C00u | (x > 0x7F800000u ? (0x0200u | ((x >> 13) & 0x03FFu)) : 0u));
        if (x >= 0x477FF000u) return (uint16_t) (sign | 0x7C00u);   // >= 65520 rounds to infinity
        if (x >= 0x38800000u) {                                     // normal range of binary16
            x -= 0x38000000u;                                       // exponent bias 127 -> 15
            x += 0x0FFFu + ((x >> 13) & 1u);                        // nearest even; may carry into the exponent
            return (uint16_t) (sign | (x >> 13));
        }
        if (x < 0x33000000u) return sign;                           // below 2^-25: zero (2^-25 itself ties to even = 0)
        const uint32_t e = x >> 23, mant = (x & 0x007FFFFFu) | 0x00800000u, shift = 126u - e;      // 14 .. 24
        uint32_t r = mant >> shift;
        const uint32_t rem = mant & ((1u << shift) - 1u), mid = 1u << (shift - 1u);
        if (rem > mid || (rem == mid && (r & 1u))) ++r;             // may reach 0x0400 = the smallest normal
        return (uint16_t) (sign | r);
    }

    /// binary16 -> float (exact; signalling NaNs are quieted)
    static float float16_to_float32(uint16_t h) {
        const uint32_t sign = (uint32_t) (h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x03FFu;
        uint32_t x;
        if (e == 0) {
            if (m == 0) {
                x = sign;
            } else {                                                // subnormal: m * 2^-24, exact in float
                float v = (float) m * 5.9604644775390625e-8f;
                memcpy(&x, &v, 4);
                x |= sign;
            }
        } else if (e == 31) {
            x = sign | 0x7F800000u | (m << 13) | (m ? 0x00400000u : 0u);
        } else {
            x = sign | ((e + 112u) << 23) | (m << 13);
        }
        float f;
        memcpy(&f, &x, 4);
        return f;
    }
};

} // namespace enoki

namespace std {
template <> struct numeric_limits<enoki::half> {
    static constexpr bool is_specialized = true;
    static constexpr bool is_signed = true, is_integer = false, is_exact = false, is_modulo = false, is_iec559 = true;
    static constexpr bool has_infinity = true, has_quiet_NaN = true, has_signaling_NaN = true, is_bounded = true;
    static constexpr int digits = 11, digits10 = 3, max_digits10 = 5, radix = 2;
    static constexpr int min_exponent = -13, min_exponent10 = -4, max_exponent = 16, max_exponent10 = 4;
    static constexpr float_denorm_style has_denorm = denorm_present;
    static constexpr float_round_style round_style = round_to_nearest;
    static enoki::half min() noexcept { return enoki::half::from_binary(0x0400); }
    static enoki::half lowest() noexcept { return enoki::half::from_binary(0xFBFF); }
    static enoki::half max() noexcept { return enoki::half::from_binary(0x7BFF); }
    static enoki::half epsilon() noexcept { return enoki::half::from_binary(0x1400); }
    static enoki::half round_error() noexcept { return enoki::half::from_binary(0x3800); }
    static enoki::half infinity() noexcept { return enoki::half::from_binary(0x7C00); }
    static enoki::half quiet_NaN() noexcept { return enoki::half::from_binary(0x7FFF); }
    static enoki::half signaling_NaN() noexcept { return enoki::half::from_binary(0x7DFF); }
    static enoki::half denorm_min() noexcept { return enoki::half::from_binary(0x0001); }
};
}
