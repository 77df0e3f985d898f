/*
    enoki/special.h -- special functions: erf, erfc, erfinv, i0e, dawson, erfi, lgamma, tgamma

    Same algorithms as the reference's math support library (include/enoki/special.h:22-312): Cephes-style
    polynomial / rational / Chebyshev approximations composed from the vertical ops of the array type, with the
    Estrin groupings of array_math.h:25-100.  Arrays that offer a fused member (`HIPArray::erf_()` ... -- one kernel,
    include/enoki/device/ek_special.h) use it; every other array type -- in particular `DiffArray`, which thereby
    differentiates through the approximation exactly like the reference does -- runs the composition below.  Both
    evaluate the same operations in the same order, so their values agree bit for bit.

    The Carlson / Legendre elliptic integrals (special.h:314-672) live in <enoki/ellint.h>, included below.
*/
#pragma once

#include <enoki/array.h>
#include <enoki/ellint.h>

#include <limits>

namespace enoki {

namespace detail {
    // ---- polyN: Estrin's scheme, coefficients rounded to the scalar type first (array_math.h:25-100) ----
    template <typename T, typename S = scalar_t<T>> inline T lit(double c) { return T(S(c)); }

    template <typename T> inline T poly2(const T &x, double c0, double c1, double c2) {
        T x2 = x * x;
        return fmadd(x2, lit<T>(c2), fmadd(x, lit<T>(c1), lit<T>(c0)));
    }
    template <typename T> inline T poly3(const T &x, double c0, double c1, double c2, double c3) {
        T x2 = x * x;
        return fmadd(x2, fmadd(x, lit<T>(c3), lit<T>(c2)), fmadd(x, lit<T>(c1), lit<T>(c0)));
    }
    template <typename T> inline T poly4(const T &x, double c0, double c1, double c2, double c3, double c4) {
        T x2 = x * x, x4 = x2 * x2;
        return fmadd(x2, fmadd(x, lit<T>(c3), lit<T>(c2)), fmadd(x, lit<T>(c1), lit<T>(c0)) + lit<T>(c4) * x4);
    }
    template <typename T> inline T poly5(const T &x, double c0, double c1, double c2, double c3, double c4, double c5) {
        T x2 = x * x, x4 = x2 * x2;
        return fmadd(x2, fmadd(x, lit<T>(c3), lit<T>(c2)),
                     fmadd(x4, fmadd(x, lit<T>(c5), lit<T>(c4)), fmadd(x, lit<T>(c1), lit<T>(c0))));
    }
    template <typename T>
    inline T poly6(const T &x, double c0, double c1, double c2, double c3, double c4, double c5, double c6) {
        T x2 = x * x, x4 = x2 * x2;
        return fmadd(x4, fmadd(x2, lit<T>(c6), fmadd(x, lit<T>(c5), lit<T>(c4))),
                     fmadd(x2, fmadd(x, lit<T>(c3), lit<T>(c2)), fmadd(x, lit<T>(c1), lit<T>(c0))));
    }
    template <typename T>
    inline T poly7(const T &x, double c0, double c1, double c2, double c3, double c4, double c5, double c6, double c7) {
        T x2 = x * x, x4 = x2 * x2;
        return fmadd(x4, fmadd(x2, fmadd(x, lit<T>(c7), lit<T>(c6)), fmadd(x, lit<T>(c5), lit<T>(c4))),
                     fmadd(x2, fmadd(x, lit<T>(c3), lit<T>(c2)), fmadd(x, lit<T>(c1), lit<T>(c0))));
    }
    template <typename T>
    inline T poly8(const T &x, double c0, double c1, double c2, double c3, double c4, double c5, double c6, double c7,
                   double c8) {
        T x2 = x * x, x4 = x2 * x2, x8 = x4 * x4;
        return fmadd(x4, fmadd(x2, fmadd(x, lit<T>(c7), lit<T>(c6)), fmadd(x, lit<T>(c5), lit<T>(c4))),
                     fmadd(x2, fmadd(x, lit<T>(c3), lit<T>(c2)), fmadd(x, lit<T>(c1), lit<T>(c0)) + lit<T>(c8) * x8));
    }

    template <typename T>
    inline T poly9(const T &x, double c0, double c1, double c2, double c3, double c4, double c5, double c6, double c7,
                   double c8, double c9) {
        T x2 = x * x, x4 = x2 * x2, x8 = x4 * x4;
        return fmadd(x8, fmadd(x, lit<T>(c9), lit<T>(c8)),
                     fmadd(x4, fmadd(x2, fmadd(x, lit<T>(c7), lit<T>(c6)), fmadd(x, lit<T>(c5), lit<T>(c4))),
                           fmadd(x2, fmadd(x, lit<T>(c3), lit<T>(c2)), fmadd(x, lit<T>(c1), lit<T>(c0)))));
    }
    template <typename T>
    inline T poly10(const T &x, double c0, double c1, double c2, double c3, double c4, double c5, double c6, double c7,
                    double c8, double c9, double c10) {
        T x2 = x * x, x4 = x2 * x2, x8 = x4 * x4;
        return fmadd(x8, fmadd(x2, lit<T>(c10), fmadd(x, lit<T>(c9), lit<T>(c8))),
                     fmadd(x4, fmadd(x2, fmadd(x, lit<T>(c7), lit<T>(c6)), fmadd(x, lit<T>(c5), lit<T>(c4))),
                           fmadd(x2, fmadd(x, lit<T>(c3), lit<T>(c2)), fmadd(x, lit<T>(c1), lit<T>(c0)))));
    }

    /// Chebyshev series at x/2 (special.h:22-36; the recurrence visits coeffs[0] twice, like the reference)
    template <typename T, size_t N> inline T chbevl(const T &x, const double (&coeffs)[N]) {
        T b0 = lit<T>(coeffs[0]), b1 = lit<T>(0), b2 = lit<T>(0);
        for (size_t i = 0; i < N; ++i) {
            b2 = b1;
            b1 = b0;
            b0 = fmsub(x, b1, b2 - lit<T>(coeffs[i]));
        }
        return (b0 - b2) * lit<T>(0.5);
    }

    template <typename T> constexpr bool is_single_v = std::is_same_v<scalar_t<T>, float>;

    template <typename T> inline T erf_core(const T &x);

    /// erfc before the |x| < 1 fix-up (special.h:56-123)
    template <typename T> inline T erfc_core(const T &x) {
        T xa = abs(x), z = exp(-x * x), r;
        if constexpr (is_single_v<T>) {
            auto large = xa > lit<T>(2);
            T q = rcp(xa), y = q * q;
            T p_small = poly8(y, 5.638259427386472e-1, -2.741127028184656e-1, 3.404879937665872e-1, -4.944515323274145e-1,
                              6.210004621745983e-1, -5.824733027278666e-1, 3.687424674597105e-1, -1.387039388740657e-1,
                              2.326819970068386e-2);
            T p_large = poly7(y, 5.641895067754075e-1, -2.820767439740514e-1, 4.218463358204948e-1, -1.015265279202700e+0,
                              2.921019019210786e+0, -7.495518717768503e+0, 1.297719955372516e+1, -1.047766399936249e+1);
            r = z * q * select(large, p_large, p_small);
        } else {
            auto large = xa > lit<T>(8);
            T p_small = poly8(xa, 5.57535335369399327526e2, 1.02755188689515710272e3, 9.34528527171957607540e2,
                              5.26445194995477358631e2, 1.96520832956077098242e2, 4.86371970985681366614e1,
                              7.46321056442269912687e0, 5.64189564831068821977e-1, 2.46196981473530512524e-10);
            T q_small = poly8(xa, 5.57535340817727675546e2, 1.65666309194161350182e3, 2.24633760818710981792e3,
                              1.82390916687909736289e3, 9.75708501743205489753e2, 3.54937778887819891062e2,
                              8.67072140885989742329e1, 1.32281951154744992508e1, 1.00000000000000000000e0);
            T p_large = poly5(xa, 2.97886665372100240670e0, 7.40974269950448939160e0, 6.16021097993053585195e0,
                              5.01905042251180477414e0, 1.27536670759978104416e0, 5.64189583547755073984e-1);
            T q_large = poly6(xa, 3.36907645100081516050e0, 9.60896809063285878198e0, 1.70814450747565897222e1,
                              1.20489539808096656605e1, 9.39603524938001434673e0, 2.26052863220117276590e0,
                              1.00000000000000000000e0);
            r = (z * select(large, p_large, p_small)) / select(large, q_large, q_small);
            r = select(neq(z, lit<T>(0)), r, lit<T>(0));
        }
        return select(x < lit<T>(0), lit<T>(2) - r, r);
    }

    /// erf before the |x| > 1 fix-up (special.h:131-156)
    template <typename T> inline T erf_core(const T &x) {
        T z = x * x, r;
        if constexpr (is_single_v<T>)
            r = poly6(z, 1.128379165726710e+0, -3.761262582423300e-1, 1.128358514861418e-1, -2.685381193529856e-2,
                      5.188327685732524e-3, -8.010193625184903e-4, 7.853861353153693e-5);
        else
            r = poly4(z, 5.55923013010394962768e4, 7.00332514112805075473e3, 2.23200534594684319226e3,
                      9.00260197203842689217e1, 9.60497373987051638749e0) /
                poly5(z, 4.92673942608635921086e4, 2.26290000613890934246e4, 4.59432382970980127987e3,
                      5.21357949780152679795e2, 3.35617141647503099647e1, 1.00000000000000000000e0);
        return r * x;
    }

    template <typename T> inline T erf_generic(const T &x) {
        return select(abs(x) > lit<T>(1), lit<T>(1) - erfc_core(x), erf_core(x));
    }
    template <typename T> inline T erfc_generic(const T &x) {
        return select(abs(x) < lit<T>(1), lit<T>(1) - erf_core(x), erfc_core(x));
    }

    template <typename T> inline T i0e_generic(const T &x_) {              // special.h:168-218
        static constexpr double A[] = { -1.30002500998624804212E-8, 6.04699502254191894932E-8,  -2.67079385394061173391E-7,
                                        1.11738753912010371815E-6,  -4.41673835845875056359E-6, 1.64484480707288970893E-5,
                                        -5.75419501008210370398E-5, 1.88502885095841655729E-4,  -5.76375574538582365885E-4,
                                        1.63947561694133579842E-3,  -4.32430999505057594430E-3, 1.05464603945949983183E-2,
                                        -2.37374148058994688156E-2, 4.93052842396707084878E-2,  -9.49010970480476444210E-2,
                                        1.71620901522208775349E-1,  -3.04682672343198398683E-1, 6.76795274409476084995E-1 };
        static constexpr double B[] = { 3.39623202570838634515E-9, 2.26666899049817806459E-8, 2.04891858946906374183E-7,
                                        2.89137052083475648297E-6, 6.88975834691682398426E-5, 3.36911647825569408990E-3,
                                        8.04490411014108831608E-1 };
        T x = abs(x_);
        T r_small = chbevl(fmsub(x, lit<T>(0.5), lit<T>(2)), A);
        T r_big = chbevl(fmsub(lit<T>(32), rcp(x), lit<T>(2)), B) * rsqrt(x);
        return select(x > lit<T>(8), r_big, r_small);
    }

    template <typename T> inline T erfinv_generic(const T &x) {            // special.h:222-246 (M. Giles)
        T w = -log((lit<T>(1) - x) * (lit<T>(1) + x));
        T w1 = w - lit<T>(2.5), w2 = sqrt(w) - lit<T>(3);
        T p1 = poly8(w1, 1.50140941, 0.246640727, -0.00417768164, -0.00125372503, 0.00021858087, -4.39150654e-06,
                     -3.5233877e-06, 3.43273939e-07, 2.81022636e-08);
        T p2 = poly8(w2, 2.83297682, 1.00167406, 0.00943887047, -0.0076224613, 0.00573950773, -0.00367342844,
                     0.00134934322, 0.000100950558, -0.000200214257);
        return select(w < lit<T>(5), p1, p2) * x;
    }

    template <typename T> inline T dawson_generic(const T &x) {            // special.h:249-265
        T x2 = x * x;
        T num = poly6(x2, 1.00000080272429, 9.18170212243285e-2, 4.25835373536124e-2, 6.0536496345054e-3,
                      9.88555033724111e-4, 3.64943550840577e-5, 1.55942290996993e-5);
        T denom = poly7(x2, 1.0, 7.58517175815194e-1, 2.81364355593059e-1, 6.81783097841267e-2, 1.13586116798019e-2,
                        1.92020805811771e-3, 5.74217664074868e-5, 3.11884331363595e-5);
        return num / denom * x;
    }

    template <typename T> inline T erfi_generic(const T &x) {              // special.h:268-272
        return lit<T>(1.12837916709551257390) * dawson_generic(x) * exp(x * x);
    }

    template <typename T> inline T lgamma_generic(const T &x_) {           // special.h:275-309 (Lanczos, g = 5, n = 6)
        static constexpr double coeff[7] = { 1.000000000190015, 76.18009172947146, -86.50532032941677, 24.01409824083091,
                                             -1.231739572450155, 0.1208650973866179e-2, -0.5395239384953e-5 };
        auto reflect = x_ < lit<T>(0.5);
        T x = select(reflect, -x_, x_ - lit<T>(1)), b = x + lit<T>(5) + lit<T>(0.5);
        T sum = lit<T>(0);
        for (int i = 6; i >= 1; --i) sum = sum + lit<T>(coeff[i]) / (x + lit<T>(double(i)));
        sum = sum + lit<T>(coeff[0]);
        T result = ((lit<T>(0.91893853320467274178) + log(sum)) - b) + log(b) * (x + lit<T>(0.5));
        const double pi = 3.14159265358979323846;
        T reflected = log(abs(lit<T>(pi) / sin(lit<T>(pi) * x_))) - result;
        result = select(reflect, reflected, result);
        return select(reflect & eq(x_, round(x_)), lit<T>(std::numeric_limits<double>::infinity()), result);
    }

    template <typename T> inline T tgamma_generic(const T &x) { return exp(lgamma_generic(x)); }   // special.h:312

#define ENOKI_HIP_SPECIAL_TRAIT(name)                                                                               \
    template <typename T, typename = void> struct has_##name : std::false_type { };                                \
    template <typename T> struct has_##name<T, std::void_t<decltype(std::declval<const T &>().name##_())>> : std::true_type { };
    ENOKI_HIP_SPECIAL_TRAIT(erf) ENOKI_HIP_SPECIAL_TRAIT(erfc) ENOKI_HIP_SPECIAL_TRAIT(erfinv) ENOKI_HIP_SPECIAL_TRAIT(i0e)
    ENOKI_HIP_SPECIAL_TRAIT(dawson) ENOKI_HIP_SPECIAL_TRAIT(erfi) ENOKI_HIP_SPECIAL_TRAIT(lgamma) ENOKI_HIP_SPECIAL_TRAIT(tgamma)
#undef ENOKI_HIP_SPECIAL_TRAIT
} // namespace detail

/// Polynomial evaluation in the reference's association (array_math.h:25-105): public like there
using detail::poly2; using detail::poly3; using detail::poly4; using detail::poly5; using detail::poly6;
using detail::poly7; using detail::poly8; using detail::poly9; using detail::poly10;

#define ENOKI_HIP_SPECIAL(name)                                                                                     \
    template <typename T, enable_if_t<is_array_v<T>> = 0> inline T name(const T &x) {                              \
        if constexpr (detail::has_##name<T>::value) return x.name##_();                                            \
        else return detail::name##_generic(x);                                                                     \
    }
ENOKI_HIP_SPECIAL(erf)        // special.h:131-165
ENOKI_HIP_SPECIAL(erfc)       // special.h:56-128
ENOKI_HIP_SPECIAL(erfinv)     // special.h:222-246
ENOKI_HIP_SPECIAL(i0e)        // special.h:168-218
ENOKI_HIP_SPECIAL(dawson)     // special.h:249-265
ENOKI_HIP_SPECIAL(erfi)       // special.h:268-272
ENOKI_HIP_SPECIAL(lgamma)     // special.h:275-309
ENOKI_HIP_SPECIAL(tgamma)     // special.h:312
#undef ENOKI_HIP_SPECIAL

} // namespace enoki
