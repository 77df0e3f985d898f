// Special functions of the reference's math support library (include/enoki/special.h), one fused kernel each.
//
// The reference composes them from traced primitives; the arithmetic below follows that composition operation for
// operation -- Estrin groupings of array_math.h:25-100 (polyN), explicit fused multiply-adds only where the reference
// calls fmadd/fmsub, coefficients rounded to the element type first -- so float32 results are bit-identical to the
// CPU path except where rcp()/rsqrt() appear (erfc / the |x| > 1 branch of erf, the x > 8 branch of i0e: parity
// class C on AVX2, exact in float64 where the reference divides).
#pragma once

#include <enoki/device/ek_math.h>

namespace ek {
namespace dev {

template <typename T> __device__ __forceinline__ T exp_t(T x) {
    if constexpr (sizeof(T) == 4) return exp_f32(x); else return exp_f64(x);
}
template <typename T> __device__ __forceinline__ T log_t(T x) {
    if constexpr (sizeof(T) == 4) return log_f32(x); else return log_f64(x);
}
template <typename T> __device__ __forceinline__ T sin_t(T x) {
    T s, c;
    if constexpr (sizeof(T) == 4) sincos_f32<true, false>(x, s, c); else sincos_f64<true, false>(x, s, c);
    return s;
}
template <typename T> __device__ __forceinline__ T sqrt_t(T x) {
    if constexpr (sizeof(T) == 4) return __builtin_sqrtf(x); else return __builtin_sqrt(x);
}
template <typename T> __device__ __forceinline__ T abs_t(T x) {
    if constexpr (sizeof(T) == 4) return __builtin_fabsf(x); else return __builtin_fabs(x);
}
template <typename T> __device__ __forceinline__ T rint_t(T x) {
    if constexpr (sizeof(T) == 4) return __builtin_rintf(x); else return __builtin_rint(x);
}

// poly4 .. poly8 (array_math.h:43-100)
template <typename T> __device__ __forceinline__ T poly4(T x, T c0, T c1, T c2, T c3, T c4) {
    T x2 = x * x, x4 = x2 * x2;
    return fma_(x2, fma_(x, c3, c2), fma_(x, c1, c0) + c4 * x4);
}
template <typename T> __device__ __forceinline__ T poly5(T x, T c0, T c1, T c2, T c3, T c4, T c5) {
    T x2 = x * x, x4 = x2 * x2;
    return fma_(x2, fma_(x, c3, c2), fma_(x4, fma_(x, c5, c4), fma_(x, c1, c0)));
}
template <typename T> __device__ __forceinline__ T poly6(T x, T c0, T c1, T c2, T c3, T c4, T c5, T c6) {
    T x2 = x * x, x4 = x2 * x2;
    return fma_(x4, fma_(x2, c6, fma_(x, c5, c4)), fma_(x2, fma_(x, c3, c2), fma_(x, c1, c0)));
}
template <typename T> __device__ __forceinline__ T poly7(T x, T c0, T c1, T c2, T c3, T c4, T c5, T c6, T c7) {
    T x2 = x * x, x4 = x2 * x2;
    return fma_(x4, fma_(x2, fma_(x, c7, c6), fma_(x, c5, c4)), fma_(x2, fma_(x, c3, c2), fma_(x, c1, c0)));
}
template <typename T> __device__ __forceinline__ T poly8(T x, T c0, T c1, T c2, T c3, T c4, T c5, T c6, T c7, T c8) {
    T x2 = x * x, x4 = x2 * x2, x8 = x4 * x4;
    return fma_(x4, fma_(x2, fma_(x, c7, c6), fma_(x, c5, c4)), fma_(x2, fma_(x, c3, c2), fma_(x, c1, c0) + c8 * x8));
}

template <typename T> __device__ __forceinline__ T erf_core(T x);
template <typename T> __device__ __forceinline__ T erfc_core(T x);

/// erfc without the |x| < 1 fix-up (special.h:56-128 with Recurse = false)
template <typename T> __device__ __forceinline__ T erfc_core(T x) {
    const T xa = abs_t(x), z = exp_t(-x * x);
    T r;
    if constexpr (sizeof(T) == 4) {
        const bool large = xa > 2.0f;
        const T q = 1.0f / xa, y = q * q;
        const T p_small = poly8<T>(y, (T) 5.638259427386472e-1, (T) -2.741127028184656e-1, (T) 3.404879937665872e-1,
                                   (T) -4.944515323274145e-1, (T) 6.210004621745983e-1, (T) -5.824733027278666e-1,
                                   (T) 3.687424674597105e-1, (T) -1.387039388740657e-1, (T) 2.326819970068386e-2);
        const T p_large = poly7<T>(y, (T) 5.641895067754075e-1, (T) -2.820767439740514e-1, (T) 4.218463358204948e-1,
                                   (T) -1.015265279202700e+0, (T) 2.921019019210786e+0, (T) -7.495518717768503e+0,
                                   (T) 1.297719955372516e+1, (T) -1.047766399936249e+1);
        r = z * q * (large ? p_large : p_small);
    } else {
        const bool large = xa > 8.0;
        const T p_small = poly8<T>(xa, 5.57535335369399327526e2, 1.02755188689515710272e3, 9.34528527171957607540e2,
                                   5.26445194995477358631e2, 1.96520832956077098242e2, 4.86371970985681366614e1,
                                   7.46321056442269912687e0, 5.64189564831068821977e-1, 2.46196981473530512524e-10);
        const T q_small = poly8<T>(xa, 5.57535340817727675546e2, 1.65666309194161350182e3, 2.24633760818710981792e3,
                                   1.82390916687909736289e3, 9.75708501743205489753e2, 3.54937778887819891062e2,
                                   8.67072140885989742329e1, 1.32281951154744992508e1, 1.00000000000000000000e0);
        const T p_large = poly5<T>(xa, 2.97886665372100240670e0, 7.40974269950448939160e0, 6.16021097993053585195e0,
                                   5.01905042251180477414e0, 1.27536670759978104416e0, 5.64189583547755073984e-1);
        const T q_large = poly6<T>(xa, 3.36907645100081516050e0, 9.60896809063285878198e0, 1.70814450747565897222e1,
                                   1.20489539808096656605e1, 9.39603524938001434673e0, 2.26052863220117276590e0,
                                   1.00000000000000000000e0);
        r = (z * (large ? p_large : p_small)) / (large ? q_large : q_small);
        if (!(z != T(0))) r = T(0);                 // r &= neq(z, 0)
    }
    if (x < T(0)) r = T(2) - r;
    return r;
}

/// erf without the |x| > 1 fix-up (special.h:131-156 with Recurse = false)
template <typename T> __device__ __forceinline__ T erf_core(T x) {
    const T z = x * x;
    T r;
    if constexpr (sizeof(T) == 4)
        r = poly6<T>(z, (T) 1.128379165726710e+0, (T) -3.761262582423300e-1, (T) 1.128358514861418e-1, (T) -2.685381193529856e-2,
                     (T) 5.188327685732524e-3, (T) -8.010193625184903e-4, (T) 7.853861353153693e-5);
    else
        r = poly4<T>(z, 5.55923013010394962768e4, 7.00332514112805075473e3, 2.23200534594684319226e3,
                     9.00260197203842689217e1, 9.60497373987051638749e0) /
            poly5<T>(z, 4.92673942608635921086e4, 2.26290000613890934246e4, 4.59432382970980127987e3,
                     5.21357949780152679795e2, 3.35617141647503099647e1, 1.00000000000000000000e0);
    return r * x;
}

template <typename T> __device__ __forceinline__ T erf_t(T x) {                 // special.h:131-165
    T r = erf_core(x);
    if (abs_t(x) > T(1)) r = T(1) - erfc_core(x);
    return r;
}

template <typename T> __device__ __forceinline__ T erfc_t(T x) {                // special.h:56-128
    T r = erfc_core(x);
    if (abs_t(x) < T(1)) r = T(1) - erf_core(x);
    return r;
}

/// Chebyshev series at x/2 (special.h:22-36); note that the recurrence starts with coeffs[0] twice, like the reference
template <typename T, int N> __device__ __forceinline__ T chbevl(T x, const double (&coeffs)[N]) {
    T b0 = (T) coeffs[0], b1 = T(0), b2 = T(0);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        b2 = b1;
        b1 = b0;
        b0 = fma_(x, b1, -(b2 - (T) coeffs[i]));        // fmsub(x, b1, b2 - c)
    }
    return (b0 - b2) * T(0.5);
}

template <typename T> __device__ __forceinline__ T i0e_t(T x_) {                // special.h:168-218
    constexpr double A[] = { -1.30002500998624804212E-8, 6.04699502254191894932E-8,  -2.67079385394061173391E-7,
                             1.11738753912010371815E-6,  -4.41673835845875056359E-6, 1.64484480707288970893E-5,
                             -5.75419501008210370398E-5, 1.88502885095841655729E-4,  -5.76375574538582365885E-4,
                             1.63947561694133579842E-3,  -4.32430999505057594430E-3, 1.05464603945949983183E-2,
                             -2.37374148058994688156E-2, 4.93052842396707084878E-2,  -9.49010970480476444210E-2,
                             1.71620901522208775349E-1,  -3.04682672343198398683E-1, 6.76795274409476084995E-1 };
    constexpr double B[] = { 3.39623202570838634515E-9, 2.26666899049817806459E-8, 2.04891858946906374183E-7,
                             2.89137052083475648297E-6, 6.88975834691682398426E-5, 3.36911647825569408990E-3,
                             8.04490411014108831608E-1 };
    const T x = abs_t(x_);
    if (x > T(8))
        return chbevl<T>(fma_(T(32), T(1) / x, -T(2)), B) * (T(1) / sqrt_t(x));
    return chbevl<T>(fma_(x, T(0.5), -T(2)), A);
}

template <typename T> __device__ __forceinline__ T erfinv_t(T x) {              // special.h:222-246 (M. Giles)
    const T w = -log_t((T(1) - x) * (T(1) + x));
    const T w1 = w - T(2.5), w2 = sqrt_t(w) - T(3);
    const T p1 = poly8<T>(w1, (T) 1.50140941, (T) 0.246640727, (T) -0.00417768164, (T) -0.00125372503, (T) 0.00021858087,
                          (T) -4.39150654e-06, (T) -3.5233877e-06, (T) 3.43273939e-07, (T) 2.81022636e-08);
    const T p2 = poly8<T>(w2, (T) 2.83297682, (T) 1.00167406, (T) 0.00943887047, (T) -0.0076224613, (T) 0.00573950773,
                          (T) -0.00367342844, (T) 0.00134934322, (T) 0.000100950558, (T) -0.000200214257);
    return (w < T(5) ? p1 : p2) * x;
}

template <typename T> __device__ __forceinline__ T dawson_t(T x) {              // special.h:249-265
    const T x2 = x * x;
    const T num = poly6<T>(x2, (T) 1.00000080272429, (T) 9.18170212243285e-2, (T) 4.25835373536124e-2, (T) 6.0536496345054e-3,
                           (T) 9.88555033724111e-4, (T) 3.64943550840577e-5, (T) 1.55942290996993e-5);
    const T denom = poly7<T>(x2, (T) 1.0, (T) 7.58517175815194e-1, (T) 2.81364355593059e-1, (T) 6.81783097841267e-2,
                             (T) 1.13586116798019e-2, (T) 1.92020805811771e-3, (T) 5.74217664074868e-5, (T) 3.11884331363595e-5);
    return num / denom * x;
}

template <typename T> __device__ __forceinline__ T erfi_t(T x) {                // special.h:268-272
    return (T) 1.12837916709551257390 * dawson_t(x) * exp_t(x * x);
}

template <typename T> __device__ __forceinline__ T lgamma_t(T x_) {             // special.h:275-309 (Lanczos, g = 5, n = 6)
    const T coeff[7] = { (T) 1.000000000190015, (T) 76.18009172947146, (T) -86.50532032941677, (T) 24.01409824083091,
                         (T) -1.231739572450155, (T) 0.1208650973866179e-2, (T) -0.5395239384953e-5 };
    const T g = T(5), log_sqrt2pi = (T) 0.91893853320467274178, pi = (T) 3.14159265358979323846;
    const bool reflect = x_ < T(0.5);
    const T x = reflect ? -x_ : x_ - T(1), b = x + g + T(0.5);
    T sum = T(0);
#pragma unroll
    for (int i = 6; i >= 1; --i) sum += coeff[i] / (x + T(i));
    sum += coeff[0];
    T result = ((log_sqrt2pi + log_t(sum)) - b) + log_t(b) * (x + T(0.5));
    if (reflect) {
        result = log_t(abs_t(pi / sin_t(pi * x_))) - result;
        if (x_ == rint_t(x_)) result = __builtin_inff();
    }
    return result;
}

template <typename T> __device__ __forceinline__ T tgamma_t(T x) { return exp_t(lgamma_t(x)); }   // special.h:312

} // namespace dev
} // namespace ek
