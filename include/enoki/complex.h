/*
    enoki/complex.h -- complex numbers over array types

    Complex<Value> is {re, im} = Array<Value, 2>, so Complex<HIPArray<float>> is a pair of device arrays and every
    operation processes as many numbers as the arrays have entries.  Operation order follows the reference
    (include/enoki/complex.h:60-260): the product is `fmaddsub(re0, z1, im0 * swap(z1))`, i.e.
    re = fmsub(re0, re1, im0 * im1), im = fmadd(re0, im1, im0 * re1); division multiplies by rcp(z); exp / log / pow /
    sqrt / sin / cos compose the real functions the same way, so results match the CPU path bit for bit wherever the
    real functions do (rcp, rsqrt and functions built on them are parity class C in float32).

    Provided: real, imag, conj, squared_norm, abs, arg, rcp, + - * / (complex and real operands), exp, log, pow, sqrt,
    sin, cos, sincos, tan.  The inverse trigonometric / hyperbolic functions (complex.h:203-260) are not provided.
*/
#pragma once

#include <enoki/array.h>

#include <utility>

namespace enoki {

template <typename Value_> struct Complex : Array<Value_, 2> {
    using Value = Value_;
    using Base = Array<Value_, 2>;
    static constexpr bool IsComplex = true;

    Complex() = default;
    Complex(const Base &b) : Base(b) { }
    /// A real number (complex.h:45-47: imaginary part zero)
    Complex(const Value &re) : Base(re, Value(scalar_t<Value>(0))) { }
    Complex(const Value &re, const Value &im) : Base(re, im) { }
};

template <typename T> constexpr bool is_complex_v = false;
template <typename V> constexpr bool is_complex_v<Complex<V>> = true;

template <typename V> inline V real(const Complex<V> &z) { return z.coeff(0); }
template <typename V> inline V imag(const Complex<V> &z) { return z.coeff(1); }
template <typename V> inline V squared_norm(const Complex<V> &z) { return fmadd(imag(z), imag(z), real(z) * real(z)); }
template <typename V> inline V norm(const Complex<V> &z) { return sqrt(squared_norm(z)); }
template <typename V> inline V abs(const Complex<V> &z) { return norm(z); }
template <typename V> inline V arg(const Complex<V> &z) { return atan2(imag(z), real(z)); }
template <typename V> inline Complex<V> conj(const Complex<V> &z) { return Complex<V>(real(z), -imag(z)); }

template <typename V> inline Complex<V> rcp(const Complex<V> &z) {
    V scale = rcp(squared_norm(z));
    return Complex<V>(real(z) * scale, -imag(z) * scale);
}

template <typename V> inline Complex<V> operator+(const Complex<V> &a, const Complex<V> &b) {
    return Complex<V>(real(a) + real(b), imag(a) + imag(b));
}
template <typename V> inline Complex<V> operator-(const Complex<V> &a, const Complex<V> &b) {
    return Complex<V>(real(a) - real(b), imag(a) - imag(b));
}
template <typename V> inline Complex<V> operator-(const Complex<V> &a) { return Complex<V>(-real(a), -imag(a)); }

/// fmaddsub(re0, z1, im0 * swap(z1)) (complex.h:89-99)
template <typename V> inline Complex<V> operator*(const Complex<V> &a, const Complex<V> &b) {
    return Complex<V>(fmsub(real(a), real(b), imag(a) * imag(b)), fmadd(real(a), imag(b), imag(a) * real(b)));
}
template <typename V> inline Complex<V> operator*(const Complex<V> &a, const V &s) { return Complex<V>(real(a) * s, imag(a) * s); }
template <typename V> inline Complex<V> operator*(const V &s, const Complex<V> &a) { return Complex<V>(s * real(a), s * imag(a)); }
template <typename V> inline Complex<V> operator/(const Complex<V> &a, const Complex<V> &b) { return a * rcp(b); }
template <typename V> inline Complex<V> operator/(const Complex<V> &a, const V &s) { return Complex<V>(real(a) / s, imag(a) / s); }

template <typename V> inline Complex<V> exp(const Complex<V> &z) {
    V e = exp(real(z));
    auto sc = sincos(imag(z));
    return Complex<V>(e * sc.second, e * sc.first);
}
template <typename V> inline Complex<V> log(const Complex<V> &z) {
    return Complex<V>(V(scalar_t<V>(0.5)) * log(squared_norm(z)), arg(z));
}
template <typename V> inline Complex<V> pow(const Complex<V> &a, const Complex<V> &b) { return exp(log(a) * b); }

template <typename V> inline Complex<V> sqrt(const Complex<V> &z) {
    auto sc = sincos(arg(z) * V(scalar_t<V>(0.5)));
    V r = sqrt(abs(z));
    return Complex<V>(sc.second * r, sc.first * r);
}

template <typename V> inline std::pair<Complex<V>, Complex<V>> sincos(const Complex<V> &z) {
    auto sc = sincos(real(z));
    auto sch = sincosh(imag(z));
    return { Complex<V>(sc.first * sch.second, sc.second * sch.first), Complex<V>(sc.second * sch.second, -sc.first * sch.first) };
}
template <typename V> inline Complex<V> sin(const Complex<V> &z) { return sincos(z).first; }
template <typename V> inline Complex<V> cos(const Complex<V> &z) { return sincos(z).second; }
template <typename V> inline Complex<V> tan(const Complex<V> &z) {
    auto sc = sincos(z);
    return sc.first / sc.second;
}

// ---- hyperbolic functions: sinh(a + ib) = sinh a cos b + i cosh a sin b, cosh(a + ib) = cosh a cos b + i sinh a sin b
//      (reference complex.h:223-251) ----
template <typename V> inline std::pair<Complex<V>, Complex<V>> sincosh(const Complex<V> &z) {
    auto sc = sincos(imag(z));
    auto sch = sincosh(real(z));
    return { Complex<V>(sch.first * sc.second, sch.second * sc.first), Complex<V>(sch.second * sc.second, sch.first * sc.first) };
}
template <typename V> inline Complex<V> sinh(const Complex<V> &z) { return sincosh(z).first; }
template <typename V> inline Complex<V> cosh(const Complex<V> &z) { return sincosh(z).second; }
template <typename V> inline Complex<V> tanh(const Complex<V> &z) {
    auto sch = sincosh(z);
    return sch.first / sch.second;
}

// ---- inverse functions through the principal logarithm and square root (reference complex.h:202-267):
//      asin z = -i log(i z + sqrt(1 - z^2)),  acos z = -i log(z + i sqrt(1 - z^2)),  atan z = -i/2 log((i - z) / (i + z)),
//      asinh z = log(z + sqrt(z^2 + 1)),  acosh z = log(z + sqrt(z^2 - 1)),  atanh z = 1/2 log((1 + z) / (1 - z)) ----
namespace detail {
    template <typename V> inline Complex<V> complex_one() { return Complex<V>(V(scalar_t<V>(1)), V(scalar_t<V>(0))); }
    template <typename V> inline Complex<V> complex_minus_i_times(const Complex<V> &w) { return Complex<V>(imag(w), -real(w)); }
    template <typename V> inline Complex<V> complex_i_times(const Complex<V> &w) { return Complex<V>(-imag(w), real(w)); }
}
template <typename V> inline Complex<V> asin(const Complex<V> &z) {
    return detail::complex_minus_i_times(log(detail::complex_i_times(z) + sqrt(detail::complex_one<V>() - z * z)));
}
template <typename V> inline Complex<V> acos(const Complex<V> &z) {
    return detail::complex_minus_i_times(log(z + detail::complex_i_times(sqrt(detail::complex_one<V>() - z * z))));
}
template <typename V> inline Complex<V> atan(const Complex<V> &z) {
    const Complex<V> i(V(scalar_t<V>(0)), V(scalar_t<V>(1)));
    Complex<V> w = log((i - z) / (i + z));
    return Complex<V>(imag(w) * V(scalar_t<V>(0.5)), -real(w) * V(scalar_t<V>(0.5)));
}
template <typename V> inline Complex<V> asinh(const Complex<V> &z) { return log(z + sqrt(z * z + detail::complex_one<V>())); }
template <typename V> inline Complex<V> acosh(const Complex<V> &z) { return log(z + sqrt(z * z - detail::complex_one<V>())); }
template <typename V> inline Complex<V> atanh(const Complex<V> &z) {
    const Complex<V> one = detail::complex_one<V>();
    Complex<V> w = log((one + z) / (one - z));
    return Complex<V>(real(w) * V(scalar_t<V>(0.5)), imag(w) * V(scalar_t<V>(0.5)));
}

} // namespace enoki
