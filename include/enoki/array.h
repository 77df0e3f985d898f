/*
    enoki/array.h -- the slice of the Enoki free-function vocabulary that the HIP hot path needs

    This header is written from scratch for the MI355X backend.  It provides what the reference
    spreads over array_traits.h / array_router.h / array_struct.h, restricted to the device hot
    path: type traits (is_array_v, scalar_t, mask_t, expr_t, ...), operator
    and function routing `enoki::op(x)` -> `x.op_()`, gather/scatter with array targets, and a
    small static `Array<Value, N>` for SoA structures such as `Array<HIPArray<float>, 3>`.

    Routing rule (cf. the reference's ENOKI_ROUTE_* macros, array_router.h:23-149): both operands
    are converted to the "expression type" -- the operand with the larger `Rank` (nesting depth,
    then differentiability) -- and the member `op_()` of that type is called.
*/
#pragma once

#include <algorithm>
#include <array>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <ios>
#include <limits>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>

/// Annotation macros that templated code written against the reference uses (fwd.h:17-60)
#if !defined(ENOKI_INLINE)
#  define ENOKI_INLINE inline __attribute__((always_inline))
#  define ENOKI_NOINLINE __attribute__((noinline))
#  define ENOKI_LIKELY(x) __builtin_expect(!!(x), 1)
#  define ENOKI_UNLIKELY(x) __builtin_expect(!!(x), 0)
#  define ENOKI_MARK_USED(x) (void) x
#endif

namespace enoki {

// ---------------------------------------------------------------------------------------------
//  Traits
// ---------------------------------------------------------------------------------------------

/// Every array type derives from this empty tag
struct ArrayTag { };

template <bool B> using enable_if_t = std::enable_if_t<B, int>;

template <typename T> constexpr bool is_array_v = std::is_base_of_v<ArrayTag, std::decay_t<T>>;

namespace detail {
    struct reinterpret_flag { };

    template <typename T, typename = int> struct scalar { using type = std::decay_t<T>; };
    template <typename T> struct scalar<T, enable_if_t<is_array_v<T>>> { using type = typename std::decay_t<T>::Scalar; };

    template <typename T, typename = int> struct value { using type = std::decay_t<T>; };
    template <typename T> struct value<T, enable_if_t<is_array_v<T>>> { using type = typename std::decay_t<T>::Value; };

    template <typename T, typename = int> struct mask { using type = bool; };
    template <typename T> struct mask<T, enable_if_t<is_array_v<T>>> { using type = typename std::decay_t<T>::MaskType; };

    template <typename T, typename = int> struct rank { static constexpr size_t value = 0; };
    template <typename T> struct rank<T, enable_if_t<is_array_v<T>>> { static constexpr size_t value = std::decay_t<T>::Rank; };

    template <typename T, typename = int> struct depth { static constexpr size_t value = 0; };
    template <typename T> struct depth<T, enable_if_t<is_array_v<T>>> { static constexpr size_t value = std::decay_t<T>::Depth; };

    template <typename T, typename S, typename = int> struct replace_scalar { using type = S; };
    template <typename T, typename S> struct replace_scalar<T, S, enable_if_t<is_array_v<T>>> {
        using type = typename std::decay_t<T>::template ReplaceScalar<S>;
    };

    template <typename T, typename = int> struct is_mask_impl : std::is_same<std::decay_t<T>, bool> { };
    template <typename T> struct is_mask_impl<T, enable_if_t<is_array_v<T>>> : std::bool_constant<std::decay_t<T>::IsMask> { };

    template <typename T, typename = int> struct is_diff_impl : std::false_type { };
    template <typename T> struct is_diff_impl<T, enable_if_t<is_array_v<T>>> : std::bool_constant<std::decay_t<T>::IsDiff> { };

    template <typename T, typename = int> struct is_dynamic_impl : std::false_type { };
    template <typename T> struct is_dynamic_impl<T, enable_if_t<is_array_v<T>>> : std::bool_constant<std::decay_t<T>::IsDynamic> { };

    template <typename T, typename = int> struct is_device_impl : std::false_type { };
    template <typename T> struct is_device_impl<T, enable_if_t<is_array_v<T>>> : std::bool_constant<std::decay_t<T>::IsDevice> { };

    template <typename...> constexpr bool false_v = false;
}

template <typename T> using scalar_t = typename detail::scalar<T>::type;
template <typename T> using value_t = typename detail::value<T>::type;
template <typename T> using mask_t = typename detail::mask<T>::type;
template <typename T, typename S> using replace_scalar_t = typename detail::replace_scalar<T, S>::type;
template <typename T> constexpr size_t array_depth_v = detail::depth<T>::value;
template <typename T> constexpr bool is_mask_v = detail::is_mask_impl<T>::value;
template <typename T> constexpr bool is_diff_array_v = detail::is_diff_impl<T>::value;
template <typename T> constexpr bool is_dynamic_v = detail::is_dynamic_impl<T>::value;
/// true for arrays whose storage lives in GPU memory (the role of the reference's is_cuda_array_v)
template <typename T> constexpr bool is_device_array_v = detail::is_device_impl<T>::value;
template <typename T> constexpr bool is_hip_array_v = is_device_array_v<T>;
/// The reference's name for "lives on the GPU behind a deferred-evaluation backend" (array_traits.h:228-237).  Templated
/// code written against the reference asks this to decide whether to call cuda_eval(); the HIP backend is eager, so the
/// answer that keeps such code correct is `false` (is_device_array_v tells where the storage lives).
template <typename T> constexpr bool is_cuda_array_v = false;
inline void cuda_eval(bool = false) { }
inline void cuda_sync() { }

template <typename T> using int32_array_t  = replace_scalar_t<T, int32_t>;
template <typename T> using uint32_array_t = replace_scalar_t<T, uint32_t>;
template <typename T> using int64_array_t  = replace_scalar_t<T, int64_t>;
template <typename T> using uint64_array_t = replace_scalar_t<T, uint64_t>;
template <typename T> using float32_array_t = replace_scalar_t<T, float>;
template <typename T> using float64_array_t = replace_scalar_t<T, double>;
template <typename T> using int_array_t = replace_scalar_t<T,
    std::conditional_t<sizeof(scalar_t<T>) == 8, int64_t, std::conditional_t<sizeof(scalar_t<T>) == 4, int32_t, int8_t>>>;
template <typename T> using uint_array_t = replace_scalar_t<T,
    std::conditional_t<sizeof(scalar_t<T>) == 8, uint64_t, std::conditional_t<sizeof(scalar_t<T>) == 4, uint32_t, uint8_t>>>;

// ---- the remaining public names of array_traits.h (66-155, 195-263, 499-511): spelled like the reference so that
//      templated user code (SFINAE guards, result types) compiles unchanged -----------------------------------------------
template <typename... Ts> struct identity_impl;
template <typename T> struct identity_impl<T> { using type = T; };
template <typename T> using identity_t = typename identity_impl<T>::type;

template <typename T> constexpr bool is_int8_v = std::is_same_v<T, int8_t> || std::is_same_v<T, uint8_t>;
template <typename T> constexpr bool is_int16_v = std::is_same_v<T, int16_t> || std::is_same_v<T, uint16_t>;
template <typename T> constexpr bool is_int32_v = std::is_same_v<T, int32_t> || std::is_same_v<T, uint32_t>;
template <typename T> constexpr bool is_int64_v = std::is_same_v<T, int64_t> || std::is_same_v<T, uint64_t> ||
                                                  (std::is_integral_v<T> && sizeof(T) == 8 && !std::is_same_v<T, bool>);
template <typename T> constexpr bool is_float_v = std::is_same_v<T, float>;
template <typename T> constexpr bool is_double_v = std::is_same_v<T, double>;
template <typename T> constexpr bool is_std_float_v = is_float_v<T> || is_double_v<T>;
template <typename T> constexpr bool is_std_int_v = is_int32_v<T> || is_int64_v<T>;
template <typename T> constexpr bool is_std_type_v = is_std_int_v<T> || is_std_float_v<T>;
template <typename T> constexpr bool is_scalar_v = std::is_scalar_v<std::decay_t<T>>;
template <typename T> using enable_if_int32_t = enable_if_t<is_int32_v<T>>;
template <typename T> using enable_if_int64_t = enable_if_t<is_int64_v<T>>;
template <typename T> using enable_if_std_int_v = enable_if_t<is_std_int_v<T>>;
template <typename T> using enable_if_std_float_v = enable_if_t<is_std_float_v<T>>;
template <typename T> using enable_if_std_type_v = enable_if_t<is_std_type_v<T>>;

template <typename T> using is_array = std::bool_constant<is_array_v<T>>;
template <typename T> using is_mask = std::bool_constant<is_mask_v<T>>;
template <typename T> using is_diff_array = std::bool_constant<is_diff_array_v<T>>;
template <typename T> using is_cuda_array = std::bool_constant<is_cuda_array_v<T>>;
template <typename... Ts> constexpr bool is_array_any_v = (is_array_v<Ts> || ...);
/// dynamic = the size is a run-time property at the OUTERMOST level (DynamicArray, CUDAArray there; HIPArray, DiffArray here)
template <typename T> constexpr bool is_dynamic_array_v = is_array_v<T> && is_dynamic_v<T> && array_depth_v<T> == 1;
template <typename T> constexpr bool is_static_array_v = is_array_v<T> && !is_dynamic_array_v<T>;
template <typename T> using is_dynamic_array = std::bool_constant<is_dynamic_array_v<T>>;
template <typename T> using is_static_array = std::bool_constant<is_static_array_v<T>>;
template <typename T> using enable_if_array_t = enable_if_t<is_array_v<T>>;
template <typename T> using enable_if_not_array_t = enable_if_t<!is_array_v<T>>;
template <typename T> using enable_if_static_array_t = enable_if_t<is_static_array_v<T>>;
template <typename T> using enable_if_dynamic_array_t = enable_if_t<is_dynamic_array_v<T>>;
template <typename T> using enable_if_mask_t = enable_if_t<is_mask_v<T>>;
template <typename T> using enable_if_not_mask_t = enable_if_t<!is_mask_v<T>>;
template <typename T> using enable_if_diff_array_t = enable_if_t<is_diff_array_v<T>>;
template <typename T> using enable_if_cuda_t = enable_if_t<is_cuda_array_v<T>>;

template <typename T> struct array_depth { static constexpr size_t value = array_depth_v<T>; };
namespace detail {
    template <typename T, typename = int> struct array_size_impl { static constexpr size_t value = 1; };
    template <typename T> struct array_size_impl<T, enable_if_t<is_array_v<T> && !(is_dynamic_v<T> && depth<T>::value == 1)>> {
        static constexpr size_t value = std::decay_t<T>::Size;
    };
    template <typename T> struct array_size_impl<T, enable_if_t<is_array_v<T> && is_dynamic_v<T> && depth<T>::value == 1>> {
        static constexpr size_t value = size_t(-1);                 // Dynamic (array_traits.h:259-261)
    };
}
template <typename T> struct array_size : detail::array_size_impl<T> { };
template <typename T> constexpr size_t array_size_v = array_size<T>::value;
constexpr size_t Dynamic = size_t(-1);

template <typename T> using array_t = std::decay_t<T>;              // the array type behind an expression: there are no proxies here
template <typename T> using bool_array_t = replace_scalar_t<T, bool>;
template <typename T> using size_array_t = replace_scalar_t<T, size_t>;
template <typename T> using ssize_array_t = replace_scalar_t<T, std::make_signed_t<size_t>>;
template <typename T> using float_array_t = replace_scalar_t<T, std::conditional_t<sizeof(scalar_t<T>) == 8, double, float>>;

namespace detail {
    template <typename T1, typename T2> struct expr2 {
        using D1 = std::decay_t<T1>;
        using D2 = std::decay_t<T2>;
        using type = std::conditional_t<(rank<D1>::value >= rank<D2>::value), D1, D2>;
    };
}

/// Type of an expression that combines values of type T1 and T2 (the higher-ranked operand)
template <typename T1, typename T2 = T1> using expr_t = typename detail::expr2<T1, T2>::type;

namespace detail {
    /// Pass through when the type already matches, convert otherwise
    template <typename E, typename T> inline decltype(auto) as(const T &v) {
        if constexpr (std::is_same_v<E, std::decay_t<T>>)
            return (const E &) v;
        else
            return E(v);
    }
}

template <typename T> constexpr bool is_arithmetic_scalar_v = std::is_arithmetic_v<std::decay_t<T>>;

template <typename T1, typename T2>
using enable_if_array_any_t = enable_if_t<(is_array_v<T1> || is_array_v<T2>) &&
                                          (is_array_v<T1> || is_arithmetic_scalar_v<T1>) &&
                                          (is_array_v<T2> || is_arithmetic_scalar_v<T2>)>;

// ---------------------------------------------------------------------------------------------
//  Operator / function routing
// ---------------------------------------------------------------------------------------------

namespace detail {
    /// Array types that can let go of their storage handle (HIPArray, DiffArray over it): `release_expiring_()`
    template <typename T, typename = void> struct has_release_expiring : std::false_type { };
    template <typename T> struct has_release_expiring<T, std::void_t<decltype(std::declval<T &>().release_expiring_())>> : std::true_type { };
}

/// The second overload takes an EXPIRING array (a temporary, std::move(x)): once the result exists the argument lets go of its
/// handle instead of keeping it until the end of the full expression.  A C++ caller writes hsum(sin(exp(fmadd(a, x, b)))) in one
/// expression; HIPArray leaves such values unevaluated and the consumer of a chain absorbs what NOBODY ELSE holds
/// (hip.h, detail::HIPBuffer::build_chain) -- a temporary that lives on to the semicolon would count as somebody else and
/// every link would be written out (python callers never see this: their temporaries die call by call).
#define ENOKI_HIP_ROUTE_UNARY(name, member)                                                       \
    template <typename T, enable_if_t<is_array_v<T>> = 0> inline auto name(const T &a) {          \
        return a.member##_();                                                                     \
    }                                                                                             \
    template <typename T, enable_if_t<!std::is_reference_v<T> && !std::is_const_v<T> && is_array_v<T> && \
                                      detail::has_release_expiring<T>::value> = 0>                \
    inline auto name(T &&a) {                                                                     \
        auto result = a.member##_();                                                              \
        a.release_expiring_();                                                                    \
        return result;                                                                            \
    }

#define ENOKI_HIP_ROUTE_BINARY(name, member)                                                      \
    template <typename T1, typename T2, enable_if_array_any_t<T1, T2> = 0>                        \
    inline auto name(const T1 &a1, const T2 &a2) {                                                \
        using E = expr_t<T1, T2>;                                                                 \
        return detail::as<E>(a1).member##_(detail::as<E>(a2));                                    \
    }

#define ENOKI_HIP_ROUTE_TERNARY(name, member)                                                     \
    template <typename T1, typename T2, typename T3,                                              \
              enable_if_t<is_array_v<T1> || is_array_v<T2> || is_array_v<T3>> = 0>                \
    inline auto name(const T1 &a1, const T2 &a2, const T3 &a3) {                                  \
        using E = expr_t<expr_t<T1, T2>, T3>;                                                     \
        return detail::as<E>(a1).member##_(detail::as<E>(a2), detail::as<E>(a3));                 \
    }

ENOKI_HIP_ROUTE_UNARY(operator-, neg)
ENOKI_HIP_ROUTE_UNARY(operator~, not)
ENOKI_HIP_ROUTE_UNARY(operator!, not)

ENOKI_HIP_ROUTE_BINARY(operator+, add)
ENOKI_HIP_ROUTE_BINARY(operator-, sub)
ENOKI_HIP_ROUTE_BINARY(operator*, mul)
ENOKI_HIP_ROUTE_BINARY(operator/, div)
ENOKI_HIP_ROUTE_BINARY(operator%, mod)
ENOKI_HIP_ROUTE_BINARY(operator<<, sl)
ENOKI_HIP_ROUTE_BINARY(operator>>, sr)
ENOKI_HIP_ROUTE_BINARY(operator<, lt)
ENOKI_HIP_ROUTE_BINARY(operator<=, le)
ENOKI_HIP_ROUTE_BINARY(operator>, gt)
ENOKI_HIP_ROUTE_BINARY(operator>=, ge)
ENOKI_HIP_ROUTE_BINARY(eq, eq)
ENOKI_HIP_ROUTE_BINARY(neq, neq)
ENOKI_HIP_ROUTE_BINARY(min, min)
ENOKI_HIP_ROUTE_BINARY(max, max)
ENOKI_HIP_ROUTE_BINARY(mulhi, mulhi)

ENOKI_HIP_ROUTE_TERNARY(fmadd, fmadd)
ENOKI_HIP_ROUTE_TERNARY(fmsub, fmsub)
ENOKI_HIP_ROUTE_TERNARY(fnmadd, fnmadd)
ENOKI_HIP_ROUTE_TERNARY(fnmsub, fnmsub)

ENOKI_HIP_ROUTE_UNARY(abs, abs)
ENOKI_HIP_ROUTE_UNARY(sqrt, sqrt)
ENOKI_HIP_ROUTE_UNARY(rcp, rcp)
ENOKI_HIP_ROUTE_UNARY(rsqrt, rsqrt)
ENOKI_HIP_ROUTE_UNARY(floor, floor)
ENOKI_HIP_ROUTE_UNARY(ceil, ceil)
ENOKI_HIP_ROUTE_UNARY(round, round)
ENOKI_HIP_ROUTE_UNARY(trunc, trunc)
ENOKI_HIP_ROUTE_UNARY(sin, sin)
ENOKI_HIP_ROUTE_UNARY(cos, cos)
ENOKI_HIP_ROUTE_UNARY(sincos, sincos)
ENOKI_HIP_ROUTE_UNARY(exp, exp)
ENOKI_HIP_ROUTE_UNARY(log, log)
ENOKI_HIP_ROUTE_UNARY(tan, tan)
ENOKI_HIP_ROUTE_UNARY(cot, cot)
ENOKI_HIP_ROUTE_UNARY(asin, asin)
ENOKI_HIP_ROUTE_UNARY(acos, acos)
ENOKI_HIP_ROUTE_UNARY(atan, atan)
ENOKI_HIP_ROUTE_UNARY(sinh, sinh)
ENOKI_HIP_ROUTE_UNARY(cosh, cosh)
ENOKI_HIP_ROUTE_UNARY(sincosh, sincosh)
ENOKI_HIP_ROUTE_UNARY(tanh, tanh)
ENOKI_HIP_ROUTE_UNARY(asinh, asinh)
ENOKI_HIP_ROUTE_UNARY(acosh, acosh)
ENOKI_HIP_ROUTE_UNARY(atanh, atanh)
ENOKI_HIP_ROUTE_UNARY(cbrt, cbrt)
ENOKI_HIP_ROUTE_BINARY(atan2, atan2)
ENOKI_HIP_ROUTE_BINARY(ldexp, ldexp)
ENOKI_HIP_ROUTE_UNARY(popcnt, popcnt)
ENOKI_HIP_ROUTE_UNARY(lzcnt, lzcnt)
ENOKI_HIP_ROUTE_UNARY(tzcnt, tzcnt)
ENOKI_HIP_ROUTE_UNARY(sign, sign)
ENOKI_HIP_ROUTE_UNARY(hsum, hsum)
ENOKI_HIP_ROUTE_UNARY(hprod, hprod)
ENOKI_HIP_ROUTE_UNARY(hmin, hmin)
ENOKI_HIP_ROUTE_UNARY(hmax, hmax)
ENOKI_HIP_ROUTE_UNARY(psum, psum)
ENOKI_HIP_ROUTE_UNARY(reverse, reverse)
ENOKI_HIP_ROUTE_UNARY(all, all)
ENOKI_HIP_ROUTE_UNARY(any, any)
ENOKI_HIP_ROUTE_UNARY(count, count)

template <typename T, enable_if_t<is_array_v<T>> = 0> inline bool none(const T &a) { return !any(a); }
template <typename T, typename M, enable_if_t<is_array_v<T>> = 0> inline auto extract(const T &a, const M &mask) {
    return a.extract_(detail::as<mask_t<T>>(mask));
}
template <typename T, typename M, enable_if_t<is_array_v<T>> = 0> inline T compress(const T &a, const M &mask) {
    return a.compress_(detail::as<mask_t<T>>(mask));
}
template <typename T, enable_if_t<is_array_v<T>> = 0> inline auto sqr(const T &a) { return a * a; }

// Reciprocal trigonometric / hyperbolic functions (array_math.h:463-464, 1181-1183)
template <typename T, enable_if_t<is_array_v<T>> = 0> inline auto csc(const T &a) { return rcp(sin(a)); }
template <typename T, enable_if_t<is_array_v<T>> = 0> inline auto sec(const T &a) { return rcp(cos(a)); }
template <typename T, enable_if_t<is_array_v<T>> = 0> inline auto csch(const T &a) { return rcp(sinh(a)); }
template <typename T, enable_if_t<is_array_v<T>> = 0> inline auto sech(const T &a) { return rcp(cosh(a)); }
template <typename T, enable_if_t<is_array_v<T>> = 0> inline auto coth(const T &a) { return rcp(tanh(a)); }

// The derivative weights of tan, tanh and atan as the reference composes them (autodiff.h:532-541, 685-696, 606-616).  An array type
// that has them as ONE operation of the argument (HIPArray: EK_SEC_SQR, EK_SECH_SQR, EK_RCP_1P_SQR -- same roundings) keeps the
// weight a single unevaluated map, which a chain or a reduction applies while it loads; every other type composes.
namespace detail {
    template <typename T, typename = void> struct has_sec_sqr : std::false_type { };
    template <typename T> struct has_sec_sqr<T, std::void_t<decltype(std::declval<const T &>().sec_sqr_())>> : std::true_type { };
}
template <typename T, enable_if_t<is_array_v<T>> = 0> inline auto sec_sqr(const T &a) {
    if constexpr (detail::has_sec_sqr<T>::value) return a.sec_sqr_(); else return sqr(sec(a));
}
template <typename T, enable_if_t<is_array_v<T>> = 0> inline auto sech_sqr(const T &a) {
    if constexpr (detail::has_sec_sqr<T>::value) return a.sech_sqr_(); else return sqr(sech(a));
}
template <typename T, enable_if_t<is_array_v<T>> = 0> inline auto rcp_1p_sqr(const T &a) {
    if constexpr (detail::has_sec_sqr<T>::value) return a.rcp_1p_sqr_(); else return rcp(T(1) + sqr(a));
}

namespace detail {
    template <typename T, typename = void> struct has_pow : std::false_type { };
    template <typename T> struct has_pow<T, std::void_t<decltype(std::declval<const T &>().pow_(std::declval<const T &>()))>>
        : std::true_type { };
    template <typename T, typename = void> struct has_fmod : std::false_type { };
    template <typename T> struct has_fmod<T, std::void_t<decltype(std::declval<const T &>().fmod_(std::declval<const T &>()))>>
        : std::true_type { };
}

/// pow(x, y) = exp(log(x) * y) (array_math.h:956-958): a fused kernel where the array type has one, the
/// composition (which is what a DiffArray records on its tape) otherwise
template <typename T1, typename T2, enable_if_array_any_t<T1, T2> = 0,
          enable_if_t<!std::is_integral_v<T2>> = 0>
inline auto pow(const T1 &a1, const T2 &a2) {
    using E = expr_t<T1, T2>;
    if constexpr (detail::has_pow<E>::value)
        return detail::as<E>(a1).pow_(detail::as<E>(a2));
    else
        return exp(log(detail::as<E>(a1)) * detail::as<E>(a2));
}

/// Integer powers by repeated squaring (array_math.h:960-973)
template <typename T, enable_if_t<is_array_v<T>> = 0> inline T pow(const T &x_, int y) {
    int n = y < 0 ? -y : y;
    T result(1.f), x(x_);
    while (n > 0) {
        if (n & 1) result = result * x;
        x = x * x;
        n /= 2;
    }
    return y >= 0 ? result : rcp(result);
}

/// fmod(x, y) = fnmadd(trunc(x / y), y, x) (array_math.h:1381-1383)
template <typename T1, typename T2, enable_if_array_any_t<T1, T2> = 0>
inline auto fmod(const T1 &a1, const T2 &a2) {
    using E = expr_t<T1, T2>;
    if constexpr (detail::has_fmod<E>::value)
        return detail::as<E>(a1).fmod_(detail::as<E>(a2));
    else
        return fnmadd(trunc(detail::as<E>(a1) / detail::as<E>(a2)), detail::as<E>(a2), detail::as<E>(a1));
}

/// lerp / clamp / hypot (array_math.h:1350-1379)
template <typename T1, typename T2, typename T3, enable_if_t<is_array_v<T1> || is_array_v<T2> || is_array_v<T3>> = 0>
inline auto lerp(const T1 &a, const T2 &b, const T3 &t) { return fmadd(b, t, fnmadd(a, t, a)); }
template <typename T1, typename T2, typename T3, enable_if_t<is_array_v<T1> || is_array_v<T2> || is_array_v<T3>> = 0>
inline auto clamp(const T1 &value, const T2 &lo, const T3 &hi) { return max(min(value, hi), lo); }

// Scalar fallbacks so that templated code also accepts plain arithmetic types (the reference routes these through
// array_fallbacks.h).  They are TEMPLATES on purpose: where <math.h> has put a non-template ::sqrt(float) & co. into
// the global namespace, that exact match wins and there is no ambiguity under `using namespace enoki`.
namespace detail {
    template <typename... Ts> using common_fp_t = std::common_type_t<float, Ts...>;
    template <typename... Ts> constexpr bool all_arithmetic_v = (std::is_arithmetic_v<Ts> && ...);
}
template <typename T1, typename T2, typename T3, enable_if_t<detail::all_arithmetic_v<T1, T2, T3>> = 0>
inline auto fmadd(T1 a, T2 b, T3 c) { using F = detail::common_fp_t<T1, T2, T3>; return std::fma(F(a), F(b), F(c)); }
template <typename T1, typename T2, typename T3, enable_if_t<detail::all_arithmetic_v<T1, T2, T3>> = 0>
inline auto fmsub(T1 a, T2 b, T3 c) { using F = detail::common_fp_t<T1, T2, T3>; return std::fma(F(a), F(b), -F(c)); }
template <typename T1, typename T2, typename T3, enable_if_t<detail::all_arithmetic_v<T1, T2, T3>> = 0>
inline auto fnmadd(T1 a, T2 b, T3 c) { using F = detail::common_fp_t<T1, T2, T3>; return std::fma(-F(a), F(b), F(c)); }
template <typename T1, typename T2, typename T3, enable_if_t<detail::all_arithmetic_v<T1, T2, T3>> = 0>
inline auto fnmsub(T1 a, T2 b, T3 c) { using F = detail::common_fp_t<T1, T2, T3>; return std::fma(-F(a), F(b), -F(c)); }
// Inside the fused kernels of enoki/vectorize.h (device compilation, ENOKI_HIP_DEVICE_MATH) the transcendental functions
// are the device algorithms of the stand-alone kernels -- the restated CEPHES code of array_math.h -- so that a fused
// kernel and the op-by-op program agree bit for bit; on the host they are libm.
#if defined(ENOKI_HIP_DEVICE_MATH) && defined(__HIP_DEVICE_COMPILE__)
namespace detail {
    inline float dev_sin(float a) { float s, c; ek::dev::sincos_f32<true, false>(a, s, c); return s; }
    inline float dev_cos(float a) { float s, c; ek::dev::sincos_f32<false, true>(a, s, c); return c; }
    inline double dev_sin(double a) { double s, c; ek::dev::sincos_f64<true, false>(a, s, c); return s; }
    inline double dev_cos(double a) { double s, c; ek::dev::sincos_f64<false, true>(a, s, c); return c; }
    inline float dev_tan(float a) { return ek::dev::tancot_f32<true>(a); }
    inline double dev_tan(double a) { return ek::dev::tancot_f64<true>(a); }
    inline float dev_exp(float a) { return ek::dev::exp_f32(a); }     inline double dev_exp(double a) { return ek::dev::exp_f64(a); }
    inline float dev_log(float a) { return ek::dev::log_f32(a); }     inline double dev_log(double a) { return ek::dev::log_f64(a); }
    inline float dev_asin(float a) { return ek::dev::asin_f32(a); }   inline double dev_asin(double a) { return ek::dev::asin_f64(a); }
    inline float dev_acos(float a) { return ek::dev::acos_f32(a); }   inline double dev_acos(double a) { return ek::dev::acos_f64(a); }
    inline float dev_atan(float a) { return ek::dev::atan2_f32(a, 1.0f); }
    inline double dev_atan(double a) { return ek::dev::atan2_f64(a, 1.0); }
    inline float dev_sinh(float a) { return ek::dev::sinh_f32(a); }   inline double dev_sinh(double a) { return ek::dev::sinh_f64(a); }
    inline float dev_cosh(float a) { return ek::dev::cosh_f32(a); }   inline double dev_cosh(double a) { return ek::dev::cosh_f64(a); }
    inline float dev_tanh(float a) { return ek::dev::tanh_f32(a); }   inline double dev_tanh(double a) { return ek::dev::tanh_f64(a); }
    inline float dev_cbrt(float a) { return ek::dev::cbrt_f32(a); }   inline double dev_cbrt(double a) { return ek::dev::cbrt_f64(a); }
    inline float dev_cot(float a) { return ek::dev::tancot_f32<false>(a); }
    inline double dev_cot(double a) { return ek::dev::tancot_f64<false>(a); }
    inline float dev_asinh(float a) { return ek::dev::asinh_f32(a); } inline double dev_asinh(double a) { return ek::dev::asinh_f64(a); }
    inline float dev_acosh(float a) { return ek::dev::acosh_f32(a); } inline double dev_acosh(double a) { return ek::dev::acosh_f64(a); }
    inline float dev_atanh(float a) { return ek::dev::atanh_f32(a); } inline double dev_atanh(double a) { return ek::dev::atanh_f64(a); }
    inline float dev_atan2(float y, float x) { return ek::dev::atan2_f32(y, x); }
    inline double dev_atan2(double y, double x) { return ek::dev::atan2_f64(y, x); }
    inline float dev_pow(float x, float y) { return ek::dev::pow_f32(x, y); }
    inline double dev_pow(double x, double y) { return ek::dev::pow_f64(x, y); }
}
#  define ENOKI_HIP_SCALAR_MATH(name) detail::dev_##name
#else
#  define ENOKI_HIP_SCALAR_MATH(name) detail::host_math::name
namespace detail { namespace host_math {
    using std::sin; using std::cos; using std::tan; using std::exp; using std::log; using std::asin; using std::acos; using std::atan;
    using std::sinh; using std::cosh; using std::tanh; using std::cbrt; using std::asinh; using std::acosh; using std::atanh;
    using std::atan2; using std::pow;
    template <typename T> inline T cot(T a) { return T(1) / std::tan(a); }
} }
#endif
#define ENOKI_HIP_SCALAR_UNARY(name, expr)                                                        \
    template <typename T, enable_if_t<std::is_floating_point_v<T>> = 0> inline T name(T a) { return expr; }
ENOKI_HIP_SCALAR_UNARY(sqrt, std::sqrt(a))   ENOKI_HIP_SCALAR_UNARY(rsqrt, T(1) / std::sqrt(a))
ENOKI_HIP_SCALAR_UNARY(safe_sqrt, std::sqrt(a > T(0) ? a : T(0)))
ENOKI_HIP_SCALAR_UNARY(safe_rsqrt, T(1) / std::sqrt(a > T(0) ? a : T(0)))
ENOKI_HIP_SCALAR_UNARY(sin, ENOKI_HIP_SCALAR_MATH(sin)(a))     ENOKI_HIP_SCALAR_UNARY(cos, ENOKI_HIP_SCALAR_MATH(cos)(a))
ENOKI_HIP_SCALAR_UNARY(tan, ENOKI_HIP_SCALAR_MATH(tan)(a))     ENOKI_HIP_SCALAR_UNARY(exp, ENOKI_HIP_SCALAR_MATH(exp)(a))
ENOKI_HIP_SCALAR_UNARY(log, ENOKI_HIP_SCALAR_MATH(log)(a))     ENOKI_HIP_SCALAR_UNARY(asin, ENOKI_HIP_SCALAR_MATH(asin)(a))
ENOKI_HIP_SCALAR_UNARY(acos, ENOKI_HIP_SCALAR_MATH(acos)(a))   ENOKI_HIP_SCALAR_UNARY(atan, ENOKI_HIP_SCALAR_MATH(atan)(a))
ENOKI_HIP_SCALAR_UNARY(sinh, ENOKI_HIP_SCALAR_MATH(sinh)(a))   ENOKI_HIP_SCALAR_UNARY(cosh, ENOKI_HIP_SCALAR_MATH(cosh)(a))
ENOKI_HIP_SCALAR_UNARY(tanh, ENOKI_HIP_SCALAR_MATH(tanh)(a))   ENOKI_HIP_SCALAR_UNARY(cbrt, ENOKI_HIP_SCALAR_MATH(cbrt)(a))
ENOKI_HIP_SCALAR_UNARY(cot, ENOKI_HIP_SCALAR_MATH(cot)(a))     ENOKI_HIP_SCALAR_UNARY(asinh, ENOKI_HIP_SCALAR_MATH(asinh)(a))
ENOKI_HIP_SCALAR_UNARY(acosh, ENOKI_HIP_SCALAR_MATH(acosh)(a)) ENOKI_HIP_SCALAR_UNARY(atanh, ENOKI_HIP_SCALAR_MATH(atanh)(a))
ENOKI_HIP_SCALAR_UNARY(floor, std::floor(a)) ENOKI_HIP_SCALAR_UNARY(ceil, std::ceil(a))   ENOKI_HIP_SCALAR_UNARY(abs, std::fabs(a))
#undef ENOKI_HIP_SCALAR_UNARY
template <typename T, enable_if_t<std::is_floating_point_v<T>> = 0> inline std::pair<T, T> sincos(T a) {
    return { ENOKI_HIP_SCALAR_MATH(sin)(a), ENOKI_HIP_SCALAR_MATH(cos)(a) };
}
template <typename T, enable_if_t<std::is_floating_point_v<T>> = 0> inline T atan2(T y, T x) { return ENOKI_HIP_SCALAR_MATH(atan2)(y, x); }
template <typename T, enable_if_t<std::is_floating_point_v<T>> = 0> inline T pow(T x, T y) { return ENOKI_HIP_SCALAR_MATH(pow)(x, y); }
template <typename T1, typename T2, enable_if_t<detail::all_arithmetic_v<T1, T2>> = 0> inline auto min(T1 a, T2 b) {
    using C = std::common_type_t<T1, T2>; return C(b) < C(a) ? C(b) : C(a);
}
template <typename T1, typename T2, enable_if_t<detail::all_arithmetic_v<T1, T2>> = 0> inline auto max(T1 a, T2 b) {
    using C = std::common_type_t<T1, T2>; return C(b) > C(a) ? C(b) : C(a);
}
template <typename T, enable_if_t<std::is_arithmetic_v<T>> = 0> inline T sqr(T a) { return a * a; }
template <typename T, enable_if_t<std::is_floating_point_v<T>> = 0> inline bool isnan(T a) { return a != a; }
template <typename T, enable_if_t<std::is_floating_point_v<T>> = 0> inline T mulsign(T a, T b) { return std::signbit(b) ? -a : a; }
template <typename T, enable_if_t<std::is_floating_point_v<T>> = 0> inline T sign(T a) { return std::copysign(T(1), a); }
template <typename T, enable_if_t<std::is_floating_point_v<T>> = 0> inline T copysign(T a, T b) { return std::copysign(a, b); }
/// Bit counts of scalars (array_fallbacks.h)
template <typename T, enable_if_t<std::is_integral_v<T> && !std::is_same_v<T, bool>> = 0> inline T popcnt(T v) {
    return (T) __builtin_popcountll((unsigned long long) std::make_unsigned_t<T>(v));
}
template <typename T, enable_if_t<std::is_integral_v<T> && !std::is_same_v<T, bool>> = 0> inline T lzcnt(T v) {
    using U = std::make_unsigned_t<T>;
    return v == 0 ? T(sizeof(T) * 8) : T(__builtin_clzll((unsigned long long) U(v)) - (64 - (int) sizeof(T) * 8));
}
template <typename T, enable_if_t<std::is_integral_v<T> && !std::is_same_v<T, bool>> = 0> inline T tzcnt(T v) {
    return v == 0 ? T(sizeof(T) * 8) : T(__builtin_ctzll((unsigned long long) std::make_unsigned_t<T>(v)));
}
template <typename T, enable_if_t<std::is_arithmetic_v<T>> = 0> inline T rcp(T a) { return T(1) / a; }
template <typename T, enable_if_t<std::is_arithmetic_v<T>> = 0> inline T hsum(T a) { return a; }
template <typename T, enable_if_t<std::is_arithmetic_v<T>> = 0> inline bool eq(T a, T b) { return a == b; }
template <typename T, enable_if_t<std::is_arithmetic_v<T>> = 0> inline bool neq(T a, T b) { return a != b; }
template <typename T, enable_if_t<std::is_arithmetic_v<T>> = 0> inline T select(bool m, T t, T f) { return m ? t : f; }
inline bool all(bool b) { return b; }
inline bool any(bool b) { return b; }
inline bool none(bool b) { return !b; }

/// Bit-level operators: mask & mask, int & int, and value & mask (keeps the value where the mask
/// is set and zero elsewhere -- `and_(mask)` of the backend, cuda.h:559-571)
#define ENOKI_HIP_ROUTE_BITOP(name, member)                                                       \
    template <typename T1, typename T2, enable_if_array_any_t<T1, T2> = 0>                        \
    inline auto name(const T1 &a1, const T2 &a2) {                                                \
        if constexpr (is_mask_v<T2> && !is_mask_v<T1>) {                                          \
            using M = mask_t<T1>;                                                                 \
            return a1.member##_(detail::as<M>(a2));                                               \
        } else {                                                                                  \
            using E = expr_t<T1, T2>;                                                             \
            return detail::as<E>(a1).member##_(detail::as<E>(a2));                                \
        }                                                                                         \
    }

ENOKI_HIP_ROUTE_BITOP(operator&, and)
ENOKI_HIP_ROUTE_BITOP(operator&&, and)
ENOKI_HIP_ROUTE_BITOP(operator|, or)
ENOKI_HIP_ROUTE_BITOP(operator||, or)
ENOKI_HIP_ROUTE_BITOP(operator^, xor)

template <typename T1, typename T2, enable_if_array_any_t<T1, T2> = 0>
inline auto andnot(const T1 &a1, const T2 &a2) { return a1 & !a2; }

/// Logical (zero-filling) right shift, also for signed element types
template <typename E> inline E sr_logical(const E &a, const E &k) {
    using S = scalar_t<E>;
    if constexpr (std::is_signed_v<S>) {
        using U = typename E::template ReplaceValue<std::make_unsigned_t<S>>;
        const detail::reinterpret_flag flag{};
        return E(U(a, flag) >> U(k, flag), flag);
    } else {
        return a >> k;
    }
}

/// Rotations of integer arrays (array_router.h rol/ror; cuda.h composes them from shifts as well).  Shift
/// counts >= the bit width give 0 in this backend, so a rotation by 0 is the identity.
template <typename T1, typename T2, enable_if_array_any_t<T1, T2> = 0> inline auto rol(const T1 &a, const T2 &k) {
    using E = expr_t<T1, T2>;
    using S = scalar_t<E>;
    const E bits = E(S(sizeof(S) * 8)), kk = detail::as<E>(k) & E(S(sizeof(S) * 8 - 1));
    return (detail::as<E>(a) << kk) | sr_logical(detail::as<E>(a), bits - kk);
}
template <typename T1, typename T2, enable_if_array_any_t<T1, T2> = 0> inline auto ror(const T1 &a, const T2 &k) {
    using E = expr_t<T1, T2>;
    using S = scalar_t<E>;
    const E bits = E(S(sizeof(S) * 8)), kk = detail::as<E>(k) & E(S(sizeof(S) * 8 - 1));
    return sr_logical(detail::as<E>(a), kk) | (detail::as<E>(a) << (bits - kk));
}

/// floor / ceil with conversion to an integer array type (cuda.h:485-497)
template <typename Target, typename T, enable_if_t<is_array_v<T>> = 0> inline Target floor2int(const T &a) {
    return a.template floor2int_<Target>();
}
template <typename Target, typename T, enable_if_t<is_array_v<T>> = 0> inline Target ceil2int(const T &a) {
    return a.template ceil2int_<Target>();
}

#define ENOKI_HIP_ROUTE_COMPOUND(op)                                                              \
    template <typename T1, typename T2, enable_if_t<is_array_v<T1>> = 0>                          \
    inline T1 &operator op##=(T1 &a1, const T2 &a2) {                                             \
        a1 = a1 op a2;                                                                            \
        return a1;                                                                                \
    }

ENOKI_HIP_ROUTE_COMPOUND(+)
ENOKI_HIP_ROUTE_COMPOUND(-)
ENOKI_HIP_ROUTE_COMPOUND(*)
ENOKI_HIP_ROUTE_COMPOUND(/)
ENOKI_HIP_ROUTE_COMPOUND(&)
ENOKI_HIP_ROUTE_COMPOUND(|)
ENOKI_HIP_ROUTE_COMPOUND(^)
ENOKI_HIP_ROUTE_COMPOUND(<<)
ENOKI_HIP_ROUTE_COMPOUND(>>)

/// select(mask, t, f): the value operands decide the expression type (array_router.h:480-492)
template <typename M, typename T1, typename T2,
          enable_if_t<is_array_v<M> || is_array_v<T1> || is_array_v<T2>> = 0>
inline auto select(const M &m, const T1 &t, const T2 &f) {
    using E = expr_t<T1, T2>;
    if constexpr (!is_array_v<E>) {
        // scalar branches, array mask: lift the branches to the value type that matches the mask
        using E2 = typename std::decay_t<M>::template ReplaceMaskValue<E>;
        return E2::select_(m, E2(t), E2(f));
    } else {
        return E::select_(detail::as<mask_t<E>>(m), detail::as<E>(t), detail::as<E>(f));
    }
}

/// Classification and "safe" helpers (array_router.h:606-623, array_math.h:1362-1404)
template <typename T, enable_if_t<is_array_v<T>> = 0> inline auto isnan(const T &a) { return neq(a, a); }
template <typename T, enable_if_t<is_array_v<T>> = 0> inline auto isinf(const T &a) {
    return eq(abs(a), T(std::numeric_limits<scalar_t<T>>::infinity()));
}
template <typename T, enable_if_t<is_array_v<T>> = 0> inline auto isfinite(const T &a) {
    return abs(a) < T(std::numeric_limits<scalar_t<T>>::infinity());
}
template <typename T, enable_if_t<is_array_v<T>> = 0> inline T safe_sqrt(const T &a) { return sqrt(max(a, T(scalar_t<T>(0)))); }
template <typename T, enable_if_t<is_array_v<T>> = 0> inline T safe_rsqrt(const T &a) { return rsqrt(max(a, T(scalar_t<T>(0)))); }
template <typename T, enable_if_t<is_array_v<T>> = 0> inline T safe_asin(const T &a) {
    return asin(min(T(scalar_t<T>(1)), max(T(scalar_t<T>(-1)), a)));
}
template <typename T, enable_if_t<is_array_v<T>> = 0> inline T safe_acos(const T &a) {
    return acos(min(T(scalar_t<T>(1)), max(T(scalar_t<T>(-1)), a)));
}
/// hypot without intermediate overflow / underflow (array_math.h:1362-1379)
template <typename T, enable_if_t<is_array_v<T>> = 0> inline T hypot(const T &a, const T &b) {
    using S = scalar_t<T>;
    const T inf = T(std::numeric_limits<S>::infinity());
    T abs_a = abs(a), abs_b = abs(b), maxval = max(abs_a, abs_b), minval = min(abs_a, abs_b), ratio = minval / maxval;
    return select((abs_a < inf) & (abs_b < inf) & (ratio < inf), maxval * sqrt(T(S(1)) + ratio * ratio), abs_a + abs_b);
}
/// copysign(a, b): magnitude of a, sign of b (array_router.h:376-398)
template <typename T, enable_if_t<is_array_v<T>> = 0> inline T copysign(const T &a, const T &b) { return abs(a) * sign(b); }

/// mulsign(a, b) = a * sign(b), via sign-bit xor like the CPU packets (array_router.h:447)
template <typename T, enable_if_t<is_array_v<T>> = 0> inline T mulsign(const T &a, const T &b) {
    return a * sign(b);
}

// ---------------------------------------------------------------------------------------------
//  Initialization, shape
// ---------------------------------------------------------------------------------------------

namespace detail {
    template <typename V, typename I, typename = void> struct has_gather_multi : std::false_type { };
    template <typename V, typename I>
    struct has_gather_multi<V, I, std::void_t<decltype(V::template gather_multi_<2>(
        (const V *) nullptr, (V *) nullptr, std::declval<const I &>(), std::declval<const mask_t<V> &>()))>> : std::true_type { };
}

namespace detail {
    template <typename V, typename I, typename = void> struct has_gather_records : std::false_type { };
    template <typename V, typename I>
    struct has_gather_records<V, I, std::void_t<decltype(V::template gather_records_<2>(
        (const V *) nullptr, (V *) nullptr, std::declval<const I &>(), std::declval<const mask_t<V> &>()))>> : std::true_type { };
}

namespace detail {
    /// Backends that can run several scatter_adds through one index array in one pass (HIPArray::scatter_add_multi_)
    template <typename T, typename I, typename = void> struct has_scatter_add_multi : std::false_type { };
    template <typename T, typename I>
    struct has_scatter_add_multi<T, I, std::void_t<decltype(T::scatter_add_multi_(
        size_t(0), (T *const *) nullptr, (const T *const *) nullptr, (const T *const *) nullptr, std::declval<const I &>(),
        std::declval<const mask_t<T> &>()))>> : std::true_type { };
}

/// Structure-of-arrays support for user types; specialised by ENOKI_STRUCT_SUPPORT (see the end of this file)
template <typename T, typename = int> struct struct_support { static constexpr bool Defined = false; };
template <typename T> constexpr bool is_struct_v = struct_support<std::decay_t<T>>::Defined;

template <typename T> inline T zero(size_t size = 1) {
    if constexpr (is_struct_v<T>) {
        T r;
        struct_support<T>::apply(r, [&](auto &f) { f = zero<std::decay_t<decltype(f)>>(size); });
        return r;
    } else if constexpr (is_array_v<T>) {
        return T::zero_(size);
    } else {
        return T(0);
    }
}
template <typename T> inline T empty(size_t size = 1) {
    if constexpr (is_struct_v<T>) {
        T r;
        struct_support<T>::apply(r, [&](auto &f) { f = empty<std::decay_t<decltype(f)>>(size); });
        return r;
    } else if constexpr (is_array_v<T>) {
        return T::empty_(size);
    } else {
        return T();
    }
}
template <typename T> inline T full(const scalar_t<T> &value, size_t size = 1) {
    if constexpr (is_array_v<T>) return T::full_(value, size); else return T(value);
}
template <typename T> inline T arange(size_t size) { return T::arange_(0, (ptrdiff_t) size, 1); }
template <typename T> inline T arange(ptrdiff_t start, ptrdiff_t stop, ptrdiff_t step = 1) {
    return T::arange_(start, stop, step);
}
template <typename T> inline T linspace(scalar_t<T> min, scalar_t<T> max, size_t size) {
    return T::linspace_(min, max, size);
}

/// Number of "slices" (dynamic entries) of an array; 1 for scalars
template <typename T> inline size_t slices(const T &a) {
    if constexpr (is_struct_v<T>) {
        size_t result = 0;
        struct_support<T>::apply(a, [&](const auto &f) { size_t n = slices(f); if (n > result) result = n; });
        return result;
    } else if constexpr (is_array_v<T>) {
        return a.slices_();
    } else {
        return 1;
    }
}

/// Broadcast a size-1 dynamic array to `size` entries / resize
template <typename T> inline void set_slices(T &a, size_t size) {
    if constexpr (is_struct_v<T>) struct_support<T>::apply(a, [&](auto &f) { set_slices(f, size); });
    else if constexpr (is_array_v<T>) a.set_slices_(size);
}

template <typename Target, typename Source> inline Target reinterpret_array(const Source &src) {
    if constexpr (std::is_same_v<Target, Source>) {
        return src;
    } else if constexpr (is_array_v<Target>) {
        return Target(src, detail::reinterpret_flag());
    } else {
        static_assert(sizeof(Target) == sizeof(Source));
        Target t;
        memcpy(&t, &src, sizeof(Target));
        return t;
    }
}

// ---------------------------------------------------------------------------------------------
//  Gather / scatter with an array as source / target (array_struct.h:9-123)
// ---------------------------------------------------------------------------------------------

/// gather<Array>(source, index, mask): result[i] = mask[i] ? source[index[i]] : 0
template <typename Array, size_t Stride = 0, bool Packed = true, bool IsPermute = false, typename Source,
          typename Index, typename Mask = mask_t<Index>,
          enable_if_t<is_array_v<Source> && is_dynamic_v<Source>> = 0>
inline Array gather(const Source &source, const Index &index, const Mask &mask = true) {
    return Array::template gather_array_<IsPermute>(source, index, detail::as<mask_t<Index>>(mask));
}

template <size_t Stride = 0, bool Packed = true, bool IsPermute = false, typename Target, typename Value,
          typename Index, typename Mask = mask_t<Index>,
          enable_if_t<is_array_v<Target> && is_dynamic_v<Target>> = 0>
inline void scatter(Target &target, const Value &value, const Index &index, const Mask &mask = true) {
    Target::template scatter_array_<IsPermute>(target, detail::as<Target>(value), index,
                                               detail::as<mask_t<Index>>(mask));
}

template <size_t Stride = 0, bool Packed = true, bool IsPermute = false, typename Target, typename Value,
          typename Index, typename Mask = mask_t<Index>,
          enable_if_t<is_array_v<Target> && is_dynamic_v<Target>> = 0>
inline void scatter_add(Target &target, const Value &value, const Index &index, const Mask &mask = true) {
    Target::template scatter_add_array_<IsPermute>(target, detail::as<Target>(value), index,
                                                   detail::as<mask_t<Index>>(mask));
}

// ---------------------------------------------------------------------------------------------
//  Gather / scatter of STATIC arrays and scalars from raw memory (array_router.h:1075-1131): what a function that runs
//  under vectorize() uses for indirect accesses -- `mem` is then a device pointer captured by the kernel's functor.
// ---------------------------------------------------------------------------------------------
namespace detail {
    /// One record of N scalars, loaded / stored with ONE memory instruction when N * sizeof(S) is 8 or 16 and the table is
    /// aligned to it -- a random lookup costs one request per record instead of one per component
#if defined(__clang__)
    template <typename S, size_t N> struct PackedRecord {
        typedef S Vec __attribute__((ext_vector_type(N)));   // stays ONE load / store through the optimiser
        Vec v = Vec(0);
        S get(size_t k) const { return v[k]; }
        void set(size_t k, S x) { v[k] = x; }
    };
    template <typename S, size_t N> constexpr bool packed_record_v =
        std::is_arithmetic_v<S> && !std::is_same_v<S, bool> && (N * sizeof(S) == 8 || N * sizeof(S) == 16);
#else
    template <typename S, size_t N> struct PackedRecord { S v[N] = { }; S get(size_t k) const { return v[k]; } void set(size_t k, S x) { v[k] = x; } };
    template <typename S, size_t N> constexpr bool packed_record_v = false;
#endif

    template <typename Array, typename Stored, typename Index, typename Mask>
    inline Array gather_packed(const Stored *mem, const Index &index, const Mask &mask) {
        using Inner = value_t<Array>;                     // a packet: Array<S, lanes>
        using S = scalar_t<Array>;
        constexpr size_t N = Array::Size;
        Array r;
        for (size_t i = 0; i < Inner::Size; ++i) {
            bool on;
            if constexpr (is_array_v<Mask>) on = mask.coeff(i); else on = mask;
            const size_t at = (size_t) index.coeff(i) * N;
            if constexpr (packed_record_v<Stored, N>) {
                if (((uintptr_t) mem & (N * sizeof(Stored) - 1)) == 0) {
                    PackedRecord<Stored, N> rec;
                    if (on) rec = *reinterpret_cast<const PackedRecord<Stored, N> *>(mem + at);
                    for (size_t k = 0; k < N; ++k) r.coeff(k).coeff(i) = (S) rec.get(k);
                    continue;
                }
            }
            for (size_t k = 0; k < N; ++k) r.coeff(k).coeff(i) = on ? (S) mem[at + k] : S(0);
        }
        return r;
    }

    template <typename Stored, typename Value, typename Index, typename Mask>
    inline void scatter_packed(Stored *mem, const Value &value, const Index &index, const Mask &mask) {
        using Inner = value_t<Value>;
        constexpr size_t N = Value::Size;
        for (size_t i = 0; i < Inner::Size; ++i) {
            bool on;
            if constexpr (is_array_v<Mask>) on = mask.coeff(i); else on = mask;
            if (!on) continue;
            const size_t at = (size_t) index.coeff(i) * N;
            if constexpr (packed_record_v<Stored, N>) {
                if (((uintptr_t) mem & (N * sizeof(Stored) - 1)) == 0) {
                    PackedRecord<Stored, N> rec;
                    for (size_t k = 0; k < N; ++k) rec.set(k, (Stored) value.coeff(k).coeff(i));
                    *reinterpret_cast<PackedRecord<Stored, N> *>(mem + at) = rec;
                    continue;
                }
            }
            for (size_t k = 0; k < N; ++k) mem[at + k] = (Stored) value.coeff(k).coeff(i);
        }
    }
}

template <typename Array, size_t Stride = 0, typename Index, typename Mask = bool,
          enable_if_t<!is_array_v<Array> || !is_dynamic_v<Array>> = 0>
inline Array gather(const void *mem, const Index &index, const Mask &mask = true) {
    using S = scalar_t<Array>;
    using Stored = std::conditional_t<std::is_same_v<S, bool>, uint8_t, S>;
    if constexpr (!is_array_v<Array>) {
        return mask ? (Array) static_cast<const Stored *>(mem)[index] : Array(0);
    } else {
        if constexpr (std::decay_t<Array>::Depth == 2) {
            // packed records (array_router.h:1097-1107): component k of record i lives at mem[index[i] * Size + k]
            return detail::gather_packed<Array>(static_cast<const Stored *>(mem), index, mask);
        } else {
        static_assert(std::decay_t<Array>::Depth == 1, "gather(): arrays of scalars only");
        Array r;
        for (size_t i = 0; i < Array::Size; ++i) {
            bool on;
            if constexpr (is_array_v<Mask>) on = mask.coeff(i); else on = mask;
            r.coeff(i) = on ? (S) static_cast<const Stored *>(mem)[index.coeff(i)] : S(0);
        }
        return r;
        }
    }
}

template <size_t Stride = 0, typename Value, typename Index, typename Mask = bool,
          enable_if_t<!is_array_v<Value> || !is_dynamic_v<Value>> = 0>
inline void scatter(void *mem, const Value &value, const Index &index, const Mask &mask = true) {
    using S = scalar_t<Value>;
    using Stored = std::conditional_t<std::is_same_v<S, bool>, uint8_t, S>;
    if constexpr (!is_array_v<Value>) {
        if (mask) static_cast<Stored *>(mem)[index] = (Stored) value;
    } else {
        if constexpr (std::decay_t<Value>::Depth == 2) {
            detail::scatter_packed(static_cast<Stored *>(mem), value, index, mask);
        } else {
        static_assert(std::decay_t<Value>::Depth == 1, "scatter(): arrays of scalars only");
        for (size_t i = 0; i < Value::Size; ++i) {
            bool on;
            if constexpr (is_array_v<Mask>) on = mask.coeff(i); else on = mask;
            if (on) static_cast<Stored *>(mem)[index.coeff(i)] = (Stored) value.coeff(i);
        }
        }
    }
}

// ---------------------------------------------------------------------------------------------
//  load / store of STATIC arrays and scalars from consecutive memory (array_router.h:886-1015).  Nested arrays are stored
//  component after component, i.e. Array<Packet, 3> occupies 3 * Packet::Size scalars.  The aligned and unaligned spellings
//  are the same code here (the packets of this header are plain C++ arrays; there is no aligned-move instruction to
//  pick).  Masked forms leave inactive lanes zero (load) / untouched (store).  The memory type may differ from the
//  array's scalar type only through the array's own converting constructor, e.g. Array<float, 4>(load<Array<half, 4>>(p)).
// ---------------------------------------------------------------------------------------------
namespace detail {
    template <typename T, typename Mask> inline void load_into(T &dst, const scalar_t<T> *&mem, const Mask &mask) {
        if constexpr (!is_array_v<T>) {
            dst = mask ? *mem : T(0);
            ++mem;
        } else {
            for (size_t i = 0; i < std::decay_t<T>::Size; ++i) {
                if constexpr (is_array_v<Mask>) load_into(dst.coeff(i), mem, mask.coeff(i));
                else load_into(dst.coeff(i), mem, mask);
            }
        }
    }
    template <typename T, typename Mask> inline void store_from(const T &src, scalar_t<T> *&mem, const Mask &mask) {
        if constexpr (!is_array_v<T>) {
            if (mask) *mem = src;
            ++mem;
        } else {
            for (size_t i = 0; i < std::decay_t<T>::Size; ++i) {
                if constexpr (is_array_v<Mask>) store_from(src.coeff(i), mem, mask.coeff(i));
                else store_from(src.coeff(i), mem, mask);
            }
        }
    }
}

template <typename T, typename Mask = bool, enable_if_t<!is_array_v<T> || !is_dynamic_v<T>> = 0>
inline T load_unaligned(const void *mem, const Mask &mask = true) {
    T result;
    const scalar_t<T> *p = static_cast<const scalar_t<T> *>(mem);
    detail::load_into(result, p, mask);
    return result;
}
template <typename T, typename Mask = bool, enable_if_t<!is_array_v<T> || !is_dynamic_v<T>> = 0>
inline T load(const void *mem, const Mask &mask = true) { return load_unaligned<T>(mem, mask); }

template <typename T, typename Mask = bool, enable_if_t<!is_array_v<T> || !is_dynamic_v<T>> = 0>
inline void store_unaligned(void *mem, const T &value, const Mask &mask = true) {
    scalar_t<T> *p = static_cast<scalar_t<T> *>(mem);
    detail::store_from(value, p, mask);
}
template <typename T, typename Mask = bool, enable_if_t<!is_array_v<T> || !is_dynamic_v<T>> = 0>
inline void store(void *mem, const T &value, const Mask &mask = true) { store_unaligned(mem, value, mask); }

// ---------------------------------------------------------------------------------------------
//  Static arrays: Array<Value, N> -- N components stored side by side (SoA when Value is itself a
//  dynamic array, e.g. Array<HIPArray<float>, 3> = three independent device arrays).  Every
//  operation is applied component by component through the free functions above, so it costs N
//  kernel launches on the device backend.  Component order of horizontal operations follows the
//  reference's generic static array: hsum = ((c0 + c1) + c2) ..., dot = fmadd(a2, b2, fmadd(a1, b1,
//  a0 * b0)) (array_static.h:948-960).
// ---------------------------------------------------------------------------------------------
namespace detail {
    /// Component storage of Array<Value, Size>.  bool components copy ONE BY ONE: the implicit copy of a struct of bools is
    /// a byte copy, and the gfx950 code generator does not forward the 1-bit store of a comparison result to the byte load
    /// of such a copy -- every mask of a vectorize() kernel then made a round trip through scratch memory (a store, a load
    /// and an s_waitcnt vmcnt(0) that also waited for every load in flight).
    template <typename Value, size_t Size> struct ArrayStorage {
        Value v[Size];
        Value &operator[](size_t i) { return v[i]; }
        const Value &operator[](size_t i) const { return v[i]; }
    };
    template <size_t Size> struct ArrayStorage<bool, Size> {
        bool v[Size];
        ArrayStorage() = default;
        template <typename... Args, std::enable_if_t<sizeof...(Args) == Size && (std::is_same_v<Args, bool> && ...), int> = 0>
        ArrayStorage(Args... args) : v{ args... } { }
        ArrayStorage(const ArrayStorage &o) { for (size_t i = 0; i < Size; ++i) v[i] = o.v[i]; }
        ArrayStorage &operator=(const ArrayStorage &o) { for (size_t i = 0; i < Size; ++i) v[i] = o.v[i]; return *this; }
        bool &operator[](size_t i) { return v[i]; }
        const bool &operator[](size_t i) const { return v[i]; }
    };
}

/// (the default size 1 is the one-element packet that vectorize() instantiates kernels on: one slice per GPU lane)
template <typename Value_, size_t Size_ = 1> struct Array : ArrayTag {
    using Value = Value_;
    using Scalar = scalar_t<Value_>;
    using ArrayType = Array;
    using MaskType = Array<mask_t<Value_>, Size_>;
    template <typename T> using ReplaceScalar = Array<replace_scalar_t<Value_, T>, Size_>;
    template <typename T> using ReplaceValue = Array<T, Size_>;
    template <typename T> using ReplaceMaskValue = Array<replace_scalar_t<Value_, T>, Size_>;

    static constexpr size_t Size = Size_;
    static constexpr size_t Depth = array_depth_v<Value_> + 1;
    static constexpr size_t Rank = detail::rank<Value_>::value + 4;
    static constexpr bool IsMask = is_mask_v<Value_>;
    static constexpr bool IsDiff = is_diff_array_v<Value_>;
    static constexpr bool IsDynamic = is_dynamic_v<Value_>;
    static constexpr bool IsDevice = is_device_array_v<Value_>;

    Array() = default;

    /// Broadcast a scalar or an inner array to all components
    template <typename T, enable_if_t<std::is_arithmetic_v<T> || std::is_same_v<std::decay_t<T>, Value_>> = 0>
    Array(const T &v) { for (size_t i = 0; i < Size; ++i) m_data[i] = Value(v); }

    /// One initializer per component
    template <typename... Args, enable_if_t<sizeof...(Args) == Size_ && (Size_ > 1) &&
                                            (std::is_constructible_v<Value_, const Args &> && ...)> = 0>
    Array(const Args &... args) : m_data{ Value(args)... } { }

    /// Component-wise conversion
    template <typename V2, enable_if_t<!std::is_same_v<V2, Value_>> = 0>
    Array(const Array<V2, Size_> &a) { for (size_t i = 0; i < Size; ++i) m_data[i] = Value(a.coeff(i)); }

    template <typename V2> Array(const Array<V2, Size_> &a, detail::reinterpret_flag) {
        for (size_t i = 0; i < Size; ++i) m_data[i] = reinterpret_array<Value>(a.coeff(i));
    }

    Value &coeff(size_t i) { return m_data[i]; }
    const Value &coeff(size_t i) const { return m_data[i]; }
    Value &operator[](size_t i) { return m_data[i]; }
    const Value &operator[](size_t i) const { return m_data[i]; }
    template <typename M, enable_if_t<is_mask_v<M>> = 0> auto operator[](const M &mask);
    Value &x() { return m_data[0]; }
    const Value &x() const { return m_data[0]; }
    Value &y() { static_assert(Size >= 2); return m_data[1]; }
    const Value &y() const { static_assert(Size >= 2); return m_data[1]; }
    Value &z() { static_assert(Size >= 3); return m_data[2]; }
    const Value &z() const { static_assert(Size >= 3); return m_data[2]; }
    Value &w() { static_assert(Size >= 4); return m_data[3]; }
    const Value &w() const { static_assert(Size >= 4); return m_data[3]; }
    static constexpr size_t size() { return Size; }

#define ENOKI_HIP_STATIC_UNARY(name, func)                                                        \
    Array name##_() const { Array r; for (size_t i = 0; i < Size; ++i) r.m_data[i] = func(m_data[i]); return r; }
#define ENOKI_HIP_STATIC_BINARY(name, expr)                                                       \
    Array name##_(const Array &o) const {                                                         \
        Array r;                                                                                  \
        for (size_t i = 0; i < Size; ++i) { const Value &a = m_data[i], &b = o.m_data[i]; r.m_data[i] = expr; } \
        return r;                                                                                 \
    }
#define ENOKI_HIP_STATIC_COMPARE(name, expr)                                                      \
    MaskType name##_(const Array &o) const {                                                      \
        MaskType r;                                                                               \
        for (size_t i = 0; i < Size; ++i) { const Value &a = m_data[i], &b = o.m_data[i]; r.coeff(i) = expr; } \
        return r;                                                                                 \
    }

    Array neg_() const { Array r; for (size_t i = 0; i < Size; ++i) r.m_data[i] = -m_data[i]; return r; }
    Array not_() const {
        Array r;
        for (size_t i = 0; i < Size; ++i) {
            if constexpr (std::is_same_v<Value, bool>) r.m_data[i] = !m_data[i]; else r.m_data[i] = ~m_data[i];
        }
        return r;
    }
    ENOKI_HIP_STATIC_UNARY(abs, enoki::abs)
    ENOKI_HIP_STATIC_UNARY(sqrt, enoki::sqrt) ENOKI_HIP_STATIC_UNARY(rcp, enoki::rcp) ENOKI_HIP_STATIC_UNARY(rsqrt, enoki::rsqrt)
    ENOKI_HIP_STATIC_UNARY(floor, enoki::floor) ENOKI_HIP_STATIC_UNARY(ceil, enoki::ceil)
    ENOKI_HIP_STATIC_UNARY(round, enoki::round) ENOKI_HIP_STATIC_UNARY(trunc, enoki::trunc)
    ENOKI_HIP_STATIC_UNARY(sin, enoki::sin) ENOKI_HIP_STATIC_UNARY(cos, enoki::cos) ENOKI_HIP_STATIC_UNARY(exp, enoki::exp)
    ENOKI_HIP_STATIC_UNARY(log, enoki::log) ENOKI_HIP_STATIC_UNARY(sign, enoki::sign)
    ENOKI_HIP_STATIC_UNARY(tan, enoki::tan) ENOKI_HIP_STATIC_UNARY(cot, enoki::cot) ENOKI_HIP_STATIC_UNARY(asin, enoki::asin)
    ENOKI_HIP_STATIC_UNARY(acos, enoki::acos) ENOKI_HIP_STATIC_UNARY(atan, enoki::atan) ENOKI_HIP_STATIC_UNARY(sinh, enoki::sinh)
    ENOKI_HIP_STATIC_UNARY(cosh, enoki::cosh) ENOKI_HIP_STATIC_UNARY(tanh, enoki::tanh) ENOKI_HIP_STATIC_UNARY(asinh, enoki::asinh)
    ENOKI_HIP_STATIC_UNARY(acosh, enoki::acosh) ENOKI_HIP_STATIC_UNARY(atanh, enoki::atanh) ENOKI_HIP_STATIC_UNARY(cbrt, enoki::cbrt)
    ENOKI_HIP_STATIC_UNARY(popcnt, enoki::popcnt) ENOKI_HIP_STATIC_UNARY(lzcnt, enoki::lzcnt) ENOKI_HIP_STATIC_UNARY(tzcnt, enoki::tzcnt)
    std::pair<Array, Array> sincos_() const {
        Array s, c;
        for (size_t i = 0; i < Size; ++i) { auto sc = enoki::sincos(m_data[i]); s.m_data[i] = sc.first; c.m_data[i] = sc.second; }
        return { s, c };
    }
    ENOKI_HIP_STATIC_BINARY(add, a + b) ENOKI_HIP_STATIC_BINARY(sub, a - b) ENOKI_HIP_STATIC_BINARY(mul, a * b)
    ENOKI_HIP_STATIC_BINARY(div, a / b) ENOKI_HIP_STATIC_BINARY(mod, a % b) ENOKI_HIP_STATIC_BINARY(min, enoki::min(a, b))
    ENOKI_HIP_STATIC_BINARY(max, enoki::max(a, b)) ENOKI_HIP_STATIC_BINARY(and, a & b) ENOKI_HIP_STATIC_BINARY(or, a | b)
    ENOKI_HIP_STATIC_BINARY(xor, a ^ b) ENOKI_HIP_STATIC_BINARY(sl, a << b) ENOKI_HIP_STATIC_BINARY(sr, a >> b)
    ENOKI_HIP_STATIC_BINARY(atan2, enoki::atan2(a, b)) ENOKI_HIP_STATIC_BINARY(pow, enoki::pow(a, b))
    ENOKI_HIP_STATIC_COMPARE(eq, enoki::eq(a, b)) ENOKI_HIP_STATIC_COMPARE(neq, enoki::neq(a, b))
    ENOKI_HIP_STATIC_COMPARE(lt, a < b) ENOKI_HIP_STATIC_COMPARE(le, a <= b)
    ENOKI_HIP_STATIC_COMPARE(gt, a > b) ENOKI_HIP_STATIC_COMPARE(ge, a >= b)
#undef ENOKI_HIP_STATIC_UNARY
#undef ENOKI_HIP_STATIC_BINARY
#undef ENOKI_HIP_STATIC_COMPARE

    Array fmadd_(const Array &b, const Array &c) const {
        Array r; for (size_t i = 0; i < Size; ++i) r.m_data[i] = enoki::fmadd(m_data[i], b.m_data[i], c.m_data[i]); return r;
    }
    Array fmsub_(const Array &b, const Array &c) const {
        Array r; for (size_t i = 0; i < Size; ++i) r.m_data[i] = enoki::fmsub(m_data[i], b.m_data[i], c.m_data[i]); return r;
    }
    Array fnmadd_(const Array &b, const Array &c) const {
        Array r; for (size_t i = 0; i < Size; ++i) r.m_data[i] = enoki::fnmadd(m_data[i], b.m_data[i], c.m_data[i]); return r;
    }
    Array fnmsub_(const Array &b, const Array &c) const {
        Array r; for (size_t i = 0; i < Size; ++i) r.m_data[i] = enoki::fnmsub(m_data[i], b.m_data[i], c.m_data[i]); return r;
    }

    template <typename T = Value_, enable_if_t<!is_mask_v<T>> = 0> Array and_(const MaskType &m) const {
        Array r; for (size_t i = 0; i < Size; ++i) r.m_data[i] = m_data[i] & m.coeff(i); return r;
    }
    template <typename T = Value_, enable_if_t<!is_mask_v<T>> = 0> Array or_(const MaskType &m) const {
        Array r; for (size_t i = 0; i < Size; ++i) r.m_data[i] = m_data[i] | m.coeff(i); return r;
    }

    static Array select_(const MaskType &m, const Array &t, const Array &f) {
        Array r;
        for (size_t i = 0; i < Size; ++i) r.m_data[i] = enoki::select(m.coeff(i), t.m_data[i], f.m_data[i]);
        return r;
    }

    /// Horizontal operations run over the COMPONENTS (array_static.h hsum_/hprod_/dot_)
    Value hsum_() const { Value r = m_data[0]; for (size_t i = 1; i < Size; ++i) r = r + m_data[i]; return r; }
    Value hprod_() const { Value r = m_data[0]; for (size_t i = 1; i < Size; ++i) r = r * m_data[i]; return r; }
    Value hmin_() const { Value r = m_data[0]; for (size_t i = 1; i < Size; ++i) r = enoki::min(r, m_data[i]); return r; }
    Value hmax_() const { Value r = m_data[0]; for (size_t i = 1; i < Size; ++i) r = enoki::max(r, m_data[i]); return r; }
    Value dot_(const Array &o) const {
        Value r = m_data[0] * o.m_data[0];
        for (size_t i = 1; i < Size; ++i) r = enoki::fmadd(m_data[i], o.m_data[i], r);
        return r;
    }
    auto all_() const { auto r = m_data[0]; for (size_t i = 1; i < Size; ++i) r = r & m_data[i]; return r; }
    auto any_() const { auto r = m_data[0]; for (size_t i = 1; i < Size; ++i) r = r | m_data[i]; return r; }
    /// number of active components (array_static.h count_): a size_t for masks of bools, per lane for nested static masks
    auto count_() const {
        if constexpr (std::is_same_v<Value_, bool>) {
            size_t r = 0;
            for (size_t i = 0; i < Size; ++i) r += m_data[i] ? 1 : 0;
            return r;
        } else {
            static_assert(is_array_v<Value_> && !is_dynamic_v<Value_>, "count(): per-component device masks: use count(m.x()) ...");
            Array<size_t, Value_::Size> r(size_t(0));
            for (size_t i = 0; i < Size; ++i)
                for (size_t j = 0; j < Value_::Size; ++j) r.coeff(j) += m_data[i].coeff(j) ? 1 : 0;
            return r;
        }
    }

    /// Dynamic (slice) interface when the components are dynamic arrays
    size_t slices_() const { size_t n = 0; for (size_t i = 0; i < Size; ++i) n = std::max(n, slices(m_data[i])); return n; }
    void set_slices_(size_t n) { for (size_t i = 0; i < Size; ++i) set_slices(m_data[i], n); }

    static Array zero_(size_t n = 1) { Array r; for (size_t i = 0; i < Size; ++i) r.m_data[i] = zero<Value>(n); return r; }
    static Array empty_(size_t n = 1) { Array r; for (size_t i = 0; i < Size; ++i) r.m_data[i] = empty<Value>(n); return r; }
    static Array full_(const Scalar &v, size_t n = 1) { Array r; for (size_t i = 0; i < Size; ++i) r.m_data[i] = full<Value>(v, n); return r; }

    /// Component-wise gather/scatter of SoA structures (struct_support, array_struct.h:432-465)
    template <bool IsPermute, typename Index, typename Mask>
    static Array gather_array_(const Array &source, const Index &index, const Mask &mask) {
        Array r;
        if constexpr (detail::has_gather_multi<Value, Index>::value &&
                      (is_diff_array_v<Value> ? !IsPermute : !is_diff_array_v<Index>)) {
            // device components sharing one index array: one kernel instead of Size
            if (Value::template gather_multi_<Size>(source.m_data.v, r.m_data.v, index, detail::as<mask_t<Value>>(mask)))
                return r;
        }
        for (size_t i = 0; i < Size; ++i) r.m_data[i] = gather<Value, 0, true, IsPermute>(source.m_data[i], index, mask);
        return r;
    }
    template <bool IsPermute, typename Index, typename Mask>
    static void scatter_array_(Array &target, const Array &value, const Index &index, const Mask &mask) {
        for (size_t i = 0; i < Size; ++i) scatter<0, true, IsPermute>(target.m_data[i], value.m_data[i], index, mask);
    }
    template <bool IsPermute, typename Index, typename Mask>
    static void scatter_add_array_(Array &target, const Array &value, const Index &index, const Mask &mask) {
        // the components share the index / mask array: one binning pass for all of them where the backend offers it
        // (e.g. splatting RGB samples into three image planes)
        if constexpr (detail::has_scatter_add_multi<Value, Index>::value && std::is_same_v<Mask, mask_t<Value>> &&
                      Size >= 2 && Size <= 4) {
            bool uniform = true;
            for (size_t i = 0; i < Size; ++i)
                uniform = uniform && target.m_data[i].size() == target.m_data[0].size() && target.m_data[i].size() > 1;
            if (uniform) {
                Value *targets[Size];
                const Value *values[Size];
                for (size_t i = 0; i < Size; ++i) { targets[i] = &target.m_data[i]; values[i] = &value.m_data[i]; }
                Value::scatter_add_multi_(Size, targets, values, (const Value *const *) nullptr, index, mask);
                return;
            }
        }
        for (size_t i = 0; i < Size; ++i) scatter_add<0, true, IsPermute>(target.m_data[i], value.m_data[i], index, mask);
    }

private:
    detail::ArrayStorage<Value_, Size_> m_data;
};

namespace detail {
    template <typename T, typename = void> struct has_dot : std::false_type { };
    template <typename T> struct has_dot<T, std::void_t<decltype(std::declval<const T &>().dot_(std::declval<const T &>()))>>
        : std::true_type { };
}
/// dot(a, b): fmadd chain over the components of a static array, hsum(a * b) for everything else (array_base.h:165)
template <typename T, enable_if_t<is_array_v<T>> = 0> inline auto dot(const T &a, const T &b) {
    if constexpr (detail::has_dot<T>::value) return a.dot_(b); else return hsum(a * b);
}
template <typename T, enable_if_t<is_array_v<T>> = 0> inline auto squared_norm(const T &a) { return dot(a, a); }
template <typename T, enable_if_t<is_array_v<T>> = 0> inline auto norm(const T &a) { return sqrt(dot(a, a)); }
template <typename T, enable_if_t<is_array_v<T>> = 0> inline T normalize(const T &a) { return a * rsqrt(squared_norm(a)); }
template <typename V> inline Array<V, 3> cross(const Array<V, 3> &a, const Array<V, 3> &b) {
    return Array<V, 3>(fmsub(a.y(), b.z(), a.z() * b.y()), fmsub(a.z(), b.x(), a.x() * b.z()),
                       fmsub(a.x(), b.y(), a.y() * b.x()));
}

/// meshgrid for dynamic device arrays (array_utils.h:23-48): x varies fastest
template <typename T, enable_if_t<is_array_v<T> && is_dynamic_v<T> && array_depth_v<T> == 1> = 0>
inline Array<T, 2> meshgrid(const T &x, const T &y) {
    using UInt32 = uint32_array_t<T>;
    uint32_t nx = (uint32_t) slices(x), n = nx * (uint32_t) slices(y);
    UInt32 index = arange<UInt32>(n), yi = index / UInt32(nx), xi = index - yi * UInt32(nx);
    return Array<T, 2>(gather<T>(x, xi), gather<T>(y, yi));
}

// ---- variadic field lists (up to 24 fields): statement-wise and comma-separated application of a macro ----
#define ENOKI_HIP_FE_1(M, a) M(a)
#define ENOKI_HIP_FE_2(M, a, ...) M(a) ENOKI_HIP_FE_1(M, __VA_ARGS__)
#define ENOKI_HIP_FE_3(M, a, ...) M(a) ENOKI_HIP_FE_2(M, __VA_ARGS__)
#define ENOKI_HIP_FE_4(M, a, ...) M(a) ENOKI_HIP_FE_3(M, __VA_ARGS__)
#define ENOKI_HIP_FE_5(M, a, ...) M(a) ENOKI_HIP_FE_4(M, __VA_ARGS__)
#define ENOKI_HIP_FE_6(M, a, ...) M(a) ENOKI_HIP_FE_5(M, __VA_ARGS__)
#define ENOKI_HIP_FE_7(M, a, ...) M(a) ENOKI_HIP_FE_6(M, __VA_ARGS__)
#define ENOKI_HIP_FE_8(M, a, ...) M(a) ENOKI_HIP_FE_7(M, __VA_ARGS__)
#define ENOKI_HIP_FE_9(M, a, ...) M(a) ENOKI_HIP_FE_8(M, __VA_ARGS__)
#define ENOKI_HIP_FE_10(M, a, ...) M(a) ENOKI_HIP_FE_9(M, __VA_ARGS__)
#define ENOKI_HIP_FE_11(M, a, ...) M(a) ENOKI_HIP_FE_10(M, __VA_ARGS__)
#define ENOKI_HIP_FE_12(M, a, ...) M(a) ENOKI_HIP_FE_11(M, __VA_ARGS__)
#define ENOKI_HIP_FE_13(M, a, ...) M(a) ENOKI_HIP_FE_12(M, __VA_ARGS__)
#define ENOKI_HIP_FE_14(M, a, ...) M(a) ENOKI_HIP_FE_13(M, __VA_ARGS__)
#define ENOKI_HIP_FE_15(M, a, ...) M(a) ENOKI_HIP_FE_14(M, __VA_ARGS__)
#define ENOKI_HIP_FE_16(M, a, ...) M(a) ENOKI_HIP_FE_15(M, __VA_ARGS__)
#define ENOKI_HIP_FE_17(M, a, ...) M(a) ENOKI_HIP_FE_16(M, __VA_ARGS__)
#define ENOKI_HIP_FE_18(M, a, ...) M(a) ENOKI_HIP_FE_17(M, __VA_ARGS__)
#define ENOKI_HIP_FE_19(M, a, ...) M(a) ENOKI_HIP_FE_18(M, __VA_ARGS__)
#define ENOKI_HIP_FE_20(M, a, ...) M(a) ENOKI_HIP_FE_19(M, __VA_ARGS__)
#define ENOKI_HIP_FE_21(M, a, ...) M(a) ENOKI_HIP_FE_20(M, __VA_ARGS__)
#define ENOKI_HIP_FE_22(M, a, ...) M(a) ENOKI_HIP_FE_21(M, __VA_ARGS__)
#define ENOKI_HIP_FE_23(M, a, ...) M(a) ENOKI_HIP_FE_22(M, __VA_ARGS__)
#define ENOKI_HIP_FE_24(M, a, ...) M(a) ENOKI_HIP_FE_23(M, __VA_ARGS__)
#define ENOKI_HIP_FEC_1(M, a) M(a)
#define ENOKI_HIP_FEC_2(M, a, ...) M(a), ENOKI_HIP_FEC_1(M, __VA_ARGS__)
#define ENOKI_HIP_FEC_3(M, a, ...) M(a), ENOKI_HIP_FEC_2(M, __VA_ARGS__)
#define ENOKI_HIP_FEC_4(M, a, ...) M(a), ENOKI_HIP_FEC_3(M, __VA_ARGS__)
#define ENOKI_HIP_FEC_5(M, a, ...) M(a), ENOKI_HIP_FEC_4(M, __VA_ARGS__)
#define ENOKI_HIP_FEC_6(M, a, ...) M(a), ENOKI_HIP_FEC_5(M, __VA_ARGS__)
#define ENOKI_HIP_FEC_7(M, a, ...) M(a), ENOKI_HIP_FEC_6(M, __VA_ARGS__)
#define ENOKI_HIP_FEC_8(M, a, ...) M(a), ENOKI_HIP_FEC_7(M, __VA_ARGS__)
#define ENOKI_HIP_FEC_9(M, a, ...) M(a), ENOKI_HIP_FEC_8(M, __VA_ARGS__)
#define ENOKI_HIP_FEC_10(M, a, ...) M(a), ENOKI_HIP_FEC_9(M, __VA_ARGS__)
#define ENOKI_HIP_FEC_11(M, a, ...) M(a), ENOKI_HIP_FEC_10(M, __VA_ARGS__)
#define ENOKI_HIP_FEC_12(M, a, ...) M(a), ENOKI_HIP_FEC_11(M, __VA_ARGS__)
#define ENOKI_HIP_FEC_13(M, a, ...) M(a), ENOKI_HIP_FEC_12(M, __VA_ARGS__)
#define ENOKI_HIP_FEC_14(M, a, ...) M(a), ENOKI_HIP_FEC_13(M, __VA_ARGS__)
#define ENOKI_HIP_FEC_15(M, a, ...) M(a), ENOKI_HIP_FEC_14(M, __VA_ARGS__)
#define ENOKI_HIP_FEC_16(M, a, ...) M(a), ENOKI_HIP_FEC_15(M, __VA_ARGS__)
#define ENOKI_HIP_FEC_17(M, a, ...) M(a), ENOKI_HIP_FEC_16(M, __VA_ARGS__)
#define ENOKI_HIP_FEC_18(M, a, ...) M(a), ENOKI_HIP_FEC_17(M, __VA_ARGS__)
#define ENOKI_HIP_FEC_19(M, a, ...) M(a), ENOKI_HIP_FEC_18(M, __VA_ARGS__)
#define ENOKI_HIP_FEC_20(M, a, ...) M(a), ENOKI_HIP_FEC_19(M, __VA_ARGS__)
#define ENOKI_HIP_FEC_21(M, a, ...) M(a), ENOKI_HIP_FEC_20(M, __VA_ARGS__)
#define ENOKI_HIP_FEC_22(M, a, ...) M(a), ENOKI_HIP_FEC_21(M, __VA_ARGS__)
#define ENOKI_HIP_FEC_23(M, a, ...) M(a), ENOKI_HIP_FEC_22(M, __VA_ARGS__)
#define ENOKI_HIP_FEC_24(M, a, ...) M(a), ENOKI_HIP_FEC_23(M, __VA_ARGS__)
#define ENOKI_HIP_PICK(_1, _2, _3, _4, _5, _6, _7, _8, _9, _10, _11, _12, _13, _14, _15, _16, _17, _18, _19, _20, _21, _22, _23, _24, NAME, ...) NAME
#define ENOKI_HIP_FOR_EACH(M, ...) ENOKI_HIP_PICK(__VA_ARGS__, ENOKI_HIP_FE_24, ENOKI_HIP_FE_23, ENOKI_HIP_FE_22, ENOKI_HIP_FE_21, ENOKI_HIP_FE_20, ENOKI_HIP_FE_19, ENOKI_HIP_FE_18, ENOKI_HIP_FE_17, ENOKI_HIP_FE_16, ENOKI_HIP_FE_15, ENOKI_HIP_FE_14, ENOKI_HIP_FE_13, ENOKI_HIP_FE_12, ENOKI_HIP_FE_11, ENOKI_HIP_FE_10, ENOKI_HIP_FE_9, ENOKI_HIP_FE_8, ENOKI_HIP_FE_7, ENOKI_HIP_FE_6, ENOKI_HIP_FE_5, ENOKI_HIP_FE_4, ENOKI_HIP_FE_3, ENOKI_HIP_FE_2, ENOKI_HIP_FE_1)(M, __VA_ARGS__)
#define ENOKI_HIP_FOR_EACH_COMMA(M, ...) ENOKI_HIP_PICK(__VA_ARGS__, ENOKI_HIP_FEC_24, ENOKI_HIP_FEC_23, ENOKI_HIP_FEC_22, ENOKI_HIP_FEC_21, ENOKI_HIP_FEC_20, ENOKI_HIP_FEC_19, ENOKI_HIP_FEC_18, ENOKI_HIP_FEC_17, ENOKI_HIP_FEC_16, ENOKI_HIP_FEC_15, ENOKI_HIP_FEC_14, ENOKI_HIP_FEC_13, ENOKI_HIP_FEC_12, ENOKI_HIP_FEC_11, ENOKI_HIP_FEC_10, ENOKI_HIP_FEC_9, ENOKI_HIP_FEC_8, ENOKI_HIP_FEC_7, ENOKI_HIP_FEC_6, ENOKI_HIP_FEC_5, ENOKI_HIP_FEC_4, ENOKI_HIP_FEC_3, ENOKI_HIP_FEC_2, ENOKI_HIP_FEC_1)(M, __VA_ARGS__)
#define ENOKI_HIP_FIRST(a, ...) a
#define ENOKI_HIP_CAT_(a, b) a##b
#define ENOKI_HIP_CAT(a, b) ENOKI_HIP_CAT_(a, b)

/// Structure-of-arrays support for user types (array_macro.h:216-359, array_struct.h:125-430): a class template whose
/// fields are arrays declares them once with ENOKI_STRUCT (inside the class: field-wise and converting constructors /
/// assignment) and ENOKI_STRUCT_SUPPORT (at global scope: lets zero / empty / slices / set_slices / gather / scatter /
/// scatter_add / select operate on the whole structure field by field).
#define ENOKI_HIP_S_TPL(f)      typename T_##f
#define ENOKI_HIP_S_DECL(f)     T_##f &&f##_
#define ENOKI_HIP_S_INIT(f)     f(std::forward<T_##f>(f##_))
#define ENOKI_HIP_S_COPY(f)     f(value.f)
#define ENOKI_HIP_S_ASSIGN(f)   f = value.f;
#define ENOKI_HIP_S_VISIT1(f)   fn(v.f);
#define ENOKI_HIP_S_VISIT2(f)   fn(v.f, w.f);
#define ENOKI_HIP_S_VISIT3(f)   fn(v.f, w.f, u.f);

#define ENOKI_STRUCT(Struct, ...)                                                                 \
    Struct() = default;                                                                           \
    template <ENOKI_HIP_FOR_EACH_COMMA(ENOKI_HIP_S_TPL, __VA_ARGS__),                             \
              std::enable_if_t<!std::is_base_of_v<Struct, std::decay_t<                           \
                  ENOKI_HIP_CAT(T_, ENOKI_HIP_FIRST(__VA_ARGS__))>>, int> = 0>                    \
    Struct(ENOKI_HIP_FOR_EACH_COMMA(ENOKI_HIP_S_DECL, __VA_ARGS__))                               \
        : ENOKI_HIP_FOR_EACH_COMMA(ENOKI_HIP_S_INIT, __VA_ARGS__) { }                             \
    template <typename... Args_> Struct(const Struct<Args_...> &value)                            \
        : ENOKI_HIP_FOR_EACH_COMMA(ENOKI_HIP_S_COPY, __VA_ARGS__) { }                             \
    template <typename... Args_> Struct &operator=(const Struct<Args_...> &value) {               \
        ENOKI_HIP_FOR_EACH(ENOKI_HIP_S_ASSIGN, __VA_ARGS__)                                       \
        return *this;                                                                             \
    }

/// Number of dynamic (device) arrays behind a value: 1 for a dynamic array, the sum over components / FIELDS for
/// static arrays and ENOKI_STRUCT types, 0 for scalars.  vectorize() sizes its pointer tables with it.
template <typename T, typename = int> struct dynamic_leaf_count {
    static constexpr size_t value = (is_array_v<T> && is_dynamic_v<T>) ? 1 : 0;
};
template <typename V, size_t N> struct dynamic_leaf_count<Array<V, N>, int> {
    static constexpr size_t value = N * dynamic_leaf_count<V>::value;
};
template <typename T> struct dynamic_leaf_count<T, enable_if_t<is_struct_v<T>>> {
    static constexpr size_t value = struct_support<T>::leaf_count;
};
#define ENOKI_HIP_S_LEAVES(f)   + ::enoki::dynamic_leaf_count<std::decay_t<decltype(std::declval<Value &>().f)>>::value

#define ENOKI_STRUCT_SUPPORT(Struct, ...)                                                         \
    namespace enoki {                                                                             \
    template <typename... Args_> struct struct_support<Struct<Args_...>> {                        \
        static constexpr bool Defined = true;                                                     \
        using Value = Struct<Args_...>;                                                           \
        static constexpr size_t leaf_count = 0 ENOKI_HIP_FOR_EACH(ENOKI_HIP_S_LEAVES, __VA_ARGS__); \
        template <typename F> static void apply(Value &v, F &&fn) {                               \
            ENOKI_HIP_FOR_EACH(ENOKI_HIP_S_VISIT1, __VA_ARGS__)                                   \
        }                                                                                         \
        template <typename F> static void apply(const Value &v, F &&fn) {                         \
            ENOKI_HIP_FOR_EACH(ENOKI_HIP_S_VISIT1, __VA_ARGS__)                                   \
        }                                                                                         \
        template <typename V2, typename F> static void apply2(Value &v, const V2 &w, F &&fn) {    \
            ENOKI_HIP_FOR_EACH(ENOKI_HIP_S_VISIT2, __VA_ARGS__)                                   \
        }                                                                                         \
        template <typename V2, typename V3, typename F>                                           \
        static void apply3(Value &v, const V2 &w, const V3 &u, F &&fn) {                          \
            ENOKI_HIP_FOR_EACH(ENOKI_HIP_S_VISIT3, __VA_ARGS__)                                   \
        }                                                                                         \
    };                                                                                            \
    }


// Field-wise versions of the array helpers for ENOKI_STRUCT types
template <typename T, size_t Stride = 0, bool Packed = true, bool IsPermute = false, typename Index,
          typename Mask = mask_t<Index>, enable_if_t<is_struct_v<T>> = 0>
inline T gather(const T &source, const Index &index, const Mask &mask = true) {
    T r;
    struct_support<T>::apply2(r, source, [&](auto &dst, const auto &src) {
        dst = gather<std::decay_t<decltype(dst)>, 0, true, IsPermute>(src, index, mask);
    });
    return r;
}
template <size_t Stride = 0, bool Packed = true, bool IsPermute = false, typename T, typename Index,
          typename Mask = mask_t<Index>, enable_if_t<is_struct_v<T>> = 0>
inline void scatter(T &target, const T &value, const Index &index, const Mask &mask = true) {
    struct_support<T>::apply2(target, value, [&](auto &dst, const auto &src) { scatter<0, true, IsPermute>(dst, src, index, mask); });
}
template <size_t Stride = 0, bool Packed = true, bool IsPermute = false, typename T, typename Index,
          typename Mask = mask_t<Index>, enable_if_t<is_struct_v<T>> = 0>
inline void scatter_add(T &target, const T &value, const Index &index, const Mask &mask = true) {
    struct_support<T>::apply2(target, value, [&](auto &dst, const auto &src) { scatter_add<0, true, IsPermute>(dst, src, index, mask); });
}
template <typename M, typename T, enable_if_t<is_struct_v<T>> = 0> inline T select(const M &mask, const T &t, const T &f) {
    T r;
    struct_support<T>::apply3(r, t, f, [&](auto &dst, const auto &a, const auto &b) { dst = select(mask, a, b); });
    return r;
}

// ---------------------------------------------------------------------------------------------
//  Component shuffling of static arrays (array_router.h: head / tail / concat / shuffle)
// ---------------------------------------------------------------------------------------------
template <size_t K, typename V, size_t N> inline Array<V, K> head(const Array<V, N> &a) {
    static_assert(K <= N, "head<K>(): K exceeds the array size");
    Array<V, K> r;
    for (size_t i = 0; i < K; ++i) r.coeff(i) = a.coeff(i);
    return r;
}
template <size_t K, typename V, size_t N> inline Array<V, K> tail(const Array<V, N> &a) {
    static_assert(K <= N, "tail<K>(): K exceeds the array size");
    Array<V, K> r;
    for (size_t i = 0; i < K; ++i) r.coeff(i) = a.coeff(N - K + i);
    return r;
}
template <typename V, size_t N1, size_t N2> inline Array<V, N1 + N2> concat(const Array<V, N1> &a, const Array<V, N2> &b) {
    Array<V, N1 + N2> r;
    for (size_t i = 0; i < N1; ++i) r.coeff(i) = a.coeff(i);
    for (size_t i = 0; i < N2; ++i) r.coeff(N1 + i) = b.coeff(i);
    return r;
}
template <typename V, size_t N> inline Array<V, N + 1> concat(const Array<V, N> &a, const V &b) {
    Array<V, N + 1> r;
    for (size_t i = 0; i < N; ++i) r.coeff(i) = a.coeff(i);
    r.coeff(N) = b;
    return r;
}
template <size_t... Is, typename V, size_t N> inline Array<V, sizeof...(Is)> shuffle(const Array<V, N> &a) {
    static_assert(((Is < N) && ...), "shuffle<...>(): index out of range");
    Array<V, sizeof...(Is)> r;
    size_t k = 0;
    ((r.coeff(k++) = a.coeff(Is)), ...);
    return r;
}

// ---------------------------------------------------------------------------------------------
//  Fully nested horizontal operations (array_router.h:1257-1330): reduce over every dimension down to a scalar
// ---------------------------------------------------------------------------------------------
template <typename T> inline bool all_nested(const T &a) {
    if constexpr (std::is_same_v<T, bool>) return a; else return all_nested(all(a));
}
template <typename T> inline bool any_nested(const T &a) {
    if constexpr (std::is_same_v<T, bool>) return a; else return any_nested(any(a));
}
template <typename T> inline bool none_nested(const T &a) { return !any_nested(a); }
template <typename T> inline auto hsum_nested(const T &a) {
    if constexpr (!is_array_v<T>) return a;
    else if constexpr (std::decay_t<T>::Depth == 1) return hsum(a);
    else return hsum_nested(hsum(a));
}

template <typename T> inline auto hmean(const T &a) {                       // array_base.h:167-170
    if constexpr (!is_array_v<T>) return a;
    else return hsum(a) * (1.f / (float) a.size());
}
#define ENOKI_HIP_NESTED(name)                                                                       \
    template <typename T> inline auto name##_nested(const T &a) {                                   \
        if constexpr (!is_array_v<T>) return a;                                                     \
        else if constexpr (std::decay_t<T>::Depth == 1) return name(a);                             \
        else return name##_nested(name(a));                                                         \
    }
ENOKI_HIP_NESTED(hprod) ENOKI_HIP_NESTED(hmin) ENOKI_HIP_NESTED(hmax) ENOKI_HIP_NESTED(hmean)
#undef ENOKI_HIP_NESTED
template <typename T> inline auto count_nested(const T &a) {
    if constexpr (std::is_same_v<T, bool>) return (size_t) (a ? 1 : 0); else return hsum_nested(count(a));
}

// ---------------------------------------------------------------------------------------------
//  Horizontal operations over the INNERMOST dimension (array_router.h:241-249, array_static.h:743-900): a depth-1 array is
//  reduced, a nested one keeps its outer shape -- hsum_inner(Array<Packet, 3>) is an Array of three sums.
// ---------------------------------------------------------------------------------------------
#define ENOKI_HIP_INNER(name, scalar_result)                                                         \
    template <typename T> inline auto name##_inner(const T &a) {                                    \
        if constexpr (!is_array_v<T>) {                                                             \
            return scalar_result;                                                                   \
        } else if constexpr (std::decay_t<T>::Depth == 1) {                                         \
            return name(a);                                                                         \
        } else {                                                                                    \
            using Inner = decltype(name##_inner(a.coeff(0)));                                       \
            Array<Inner, std::decay_t<T>::Size> r;                                                  \
            for (size_t i = 0; i < std::decay_t<T>::Size; ++i) r.coeff(i) = name##_inner(a.coeff(i)); \
            return r;                                                                               \
        }                                                                                           \
    }
ENOKI_HIP_INNER(hsum, a) ENOKI_HIP_INNER(hprod, a) ENOKI_HIP_INNER(hmin, a) ENOKI_HIP_INNER(hmax, a) ENOKI_HIP_INNER(hmean, a)
ENOKI_HIP_INNER(psum, a) ENOKI_HIP_INNER(all, (bool) a) ENOKI_HIP_INNER(any, (bool) a)
ENOKI_HIP_INNER(count, (size_t) ((bool) a ? 1 : 0))
#undef ENOKI_HIP_INNER
template <typename T> inline auto none_inner(const T &a) { return !any_inner(a); }

// ---------------------------------------------------------------------------------------------
//  Reductions that return `Default` for device arrays instead of synchronising (array_router.h:1327-1389): generic code
//  asks `if (any_or<true>(mask))` to skip work on the CPU and simply proceeds on the device.
// ---------------------------------------------------------------------------------------------
#define ENOKI_HIP_OR(name)                                                                           \
    template <bool Default, typename T> inline bool name##_or(const T &value) {                     \
        if constexpr (is_device_array_v<T>) { (void) value; return Default; }                       \
        else return name(value);                                                                    \
    }
ENOKI_HIP_OR(any) ENOKI_HIP_OR(all) ENOKI_HIP_OR(none) ENOKI_HIP_OR(any_nested) ENOKI_HIP_OR(all_nested) ENOKI_HIP_OR(none_nested)
#undef ENOKI_HIP_OR

// ---------------------------------------------------------------------------------------------
//  Small routines of array_router.h that compose from the above (lines 341-348, 403-470, 653-655) and of
//  array_static.h (fmaddsub / fmsubadd: 490-521, rol_array / ror_array: 599-642, low / high: 650-660)
// ---------------------------------------------------------------------------------------------
template <typename T> inline auto rad_to_deg(const T &a) { return a * scalar_t<T>(180 / 3.14159265358979323846); }
template <typename T> inline auto deg_to_rad(const T &a) { return a * scalar_t<T>(3.14159265358979323846 / 180); }
template <typename T1, typename T2> inline auto abs_dot(const T1 &a, const T2 &b) { return abs(dot(a, b)); }

/// Shifts / rotations by a compile-time amount (array_router.h:253-256)
template <size_t Imm, typename T> inline auto sl(const T &a) {
    if constexpr (is_array_v<T>) return a << T(scalar_t<T>(Imm)); else return T(a << Imm);
}
template <size_t Imm, typename T> inline auto sr(const T &a) {
    if constexpr (is_array_v<T>) return a >> T(scalar_t<T>(Imm)); else return T(a >> Imm);
}
template <size_t Imm, typename T, enable_if_t<is_array_v<T>> = 0> inline auto rol(const T &a) { return rol(a, T(scalar_t<T>(Imm))); }
template <size_t Imm, typename T, enable_if_t<is_array_v<T>> = 0> inline auto ror(const T &a) { return ror(a, T(scalar_t<T>(Imm))); }

/// floor(log2(value)) for integers (array_router.h:567-570)
template <typename T> inline auto log2i(const T &value) {
    if constexpr (is_array_v<T>) return T(scalar_t<T>(sizeof(scalar_t<T>) * 8 - 1)) - lzcnt(value);
    else return T(sizeof(T) * 8 - 1) - lzcnt(value);
}

/// Division by an integer chosen at run time (array_idiv.h:150-242).  The reference precomputes a multiplier and a shift
/// for its CPU packets; a uniform divisor costs the device nothing extra, so here the type only carries the value -- the
/// quotients are the same integers either way.  `x / divisor<T>(d)`, `d(x)`, `x % divisor_ext<T>(d)`.
template <typename T> struct divisor {
    T value = T(1);
    divisor() = default;
    divisor(T d) : value(d) { }
    template <typename T2> auto operator()(const T2 &v) const {
        if constexpr (is_array_v<T2>) return v / T2(scalar_t<T2>(value)); else return T2(v / (T2) value);
    }
};
template <typename T> struct divisor_ext : divisor<T> { using divisor<T>::divisor; };
template <typename T1, typename T2> inline auto operator/(const T1 &a, const divisor<T2> &d) { return d(a); }
template <typename T1, typename T2> inline auto operator/(const T1 &a, const divisor_ext<T2> &d) { return d(a); }
template <typename T1, typename T2> inline auto operator%(const T1 &a, const divisor_ext<T2> &d) {
    if constexpr (is_array_v<T1>) return a - d(a) * T1(scalar_t<T1>(d.value)); else return T1(a - d(a) * (T1) d.value);
}

/// Shape of a (nested) array, outermost dimension first (array_struct.h:468-539): shape(Array<HIPArray<float>, 3>) = {3, n}
namespace detail {
    template <typename T> inline void extract_shape(size_t *out, const T &a) {
        if constexpr (is_array_v<T>) {
            *out = a.size();
            if constexpr (is_array_v<value_t<T>>) { if (*out > 0) extract_shape(out + 1, a.coeff(0)); }
        }
    }
    template <typename T> inline bool is_ragged(const T &a, const size_t *shape) {
        if constexpr (is_array_v<T>) {
            if (*shape != a.size()) return true;
            bool match = true;
            if constexpr (is_static_array_v<T> && is_dynamic_v<value_t<T>>)
                for (size_t i = 0; i < a.size(); ++i) match &= !is_ragged(a.coeff(i), shape + 1);
            return !match;
        } else {
            return false;
        }
    }
    template <typename T> inline void set_shape_impl(T &a, const size_t *shape) {
        if constexpr (is_array_v<T>) {
            if constexpr (is_dynamic_array_v<T>) {
                a.resize(*shape);
            } else if constexpr (is_array_v<value_t<T>>) {
                for (size_t i = 0; i < a.size(); ++i) set_shape_impl(a.coeff(i), shape + 1);
            }
        }
    }
}
template <typename T> inline std::array<size_t, array_depth_v<T>> shape(const T &a) {
    std::array<size_t, array_depth_v<T>> result{ };
    detail::extract_shape(result.data(), a);
    return result;
}
template <typename T> inline void set_shape(T &a, const std::array<size_t, array_depth_v<T>> &value) { detail::set_shape_impl(a, value.data()); }
/// do the dynamic components of a nested array disagree about their length?
template <typename T> inline bool ragged(const T &a) { auto s = shape(a); return detail::is_ragged(a, s.data()); }

/// `os << array` (array_base.h:190-237): entries grouped so that one row is one SLICE -- an Array<HIPArray<float>, 3> prints as
/// [[x0, y0, z0],\n [x1, y1, z1], ...] -- and dynamic dimensions beyond 20 entries abbreviated to the first and last five.
/// (Device arrays are read entry by entry through coeff(): meant for debugging output.)
namespace detail {
    template <typename T, size_t N> inline auto entry_at(const T &a, const size_t (&idx)[N], size_t level) {
        // by value: coeff() of a device array returns a temporary
        if constexpr (is_array_v<T>) return entry_at(a.coeff(idx[level]), idx, level + 1); else return a;
    }
    template <typename T> constexpr bool dim_is_dynamic(size_t level) {
        if constexpr (!is_array_v<T>) return false;
        else return level == 0 ? is_dynamic_array_v<T> : dim_is_dynamic<value_t<T>>(level - 1);
    }
    template <typename Stream, typename T, size_t N>
    inline void print_entries(Stream &os, const T &a, const std::array<size_t, N> &size, size_t (&idx)[N], size_t fixed) {
        if (fixed == N) {
            const auto v = entry_at(a, idx, 0);
            if constexpr (std::is_same_v<std::decay_t<decltype(v)>, bool>) os << (v ? 1 : 0); else os << v;
            return;
        }
        const size_t k = N - fixed - 1;                       // the innermost dimension varies slowest in the output
        os << "[";
        for (size_t i = 0; i < size[k]; ++i) {
            if (dim_is_dynamic<T>(k) && size[k] > 20 && i == 5) {
                os << ".. " << size[k] - 10 << " skipped ..,";
                if (k > 0) { os << "\n"; for (size_t j = 0; j <= fixed; ++j) os << " "; } else { os << " "; }
                i = size[k] - 6;
                continue;
            }
            idx[k] = i;
            print_entries(os, a, size, idx, fixed + 1);
            if (i + 1 < size[k]) {
                if (k == 0) { os << ", "; } else { os << ",\n"; for (size_t j = 0; j <= fixed; ++j) os << " "; }
            }
        }
        os << "]";
    }
}
template <typename Stream, typename T,
          enable_if_t<is_array_v<T> && std::is_base_of_v<std::ios_base, std::decay_t<Stream>>> = 0>
inline Stream &operator<<(Stream &os, const T &a) {
    if (ragged(a)) {
        os << "[ragged array]";
    } else {
        auto size = shape(a);
        size_t idx[array_depth_v<T>] = { };
        detail::print_entries(os, a, size, idx, 0);
    }
    return os;
}

/// Conflict-free read-modify-write through an index packet (array_router.h:1168-1190, array_static.h:982-991): lane after
/// lane, `func(memory[index[i]], args[i]..., mask[i])` -- two lanes that address the same entry both take effect.  For static
/// packets and scalars over raw memory (host code, or per-lane code inside vectorize() kernels); the last argument may be a
/// mask.  Device arrays use scatter_add, which is what the reference's own CUDA path offers too.
template <typename Arg, size_t Stride = sizeof(scalar_t<Arg>), typename Func, typename Index, typename... Args>
inline void transform(void *mem, const Index &index, Func &&func, const Args &... args) {
    if constexpr (is_array_v<Arg>) {
        static_assert(!is_dynamic_v<Arg>, "transform(): static packets and scalars only; device arrays: scatter_add()");
        for (size_t i = 0; i < std::decay_t<Arg>::Size; ++i)
            transform<value_t<Arg>, Stride>(mem, index.coeff(i), func, [&](const auto &a) -> decltype(auto) {
                if constexpr (is_array_v<std::decay_t<decltype(a)>>) return a.coeff(i); else return (a);
            }(args)...);
    } else {
        Arg &ref = *reinterpret_cast<Arg *>(static_cast<uint8_t *>(mem) + (size_t) index * Stride);
        if constexpr (sizeof...(Args) > 0 && (false || ... || std::is_same_v<std::decay_t<Args>, bool>)) {
            if ((... , (bool) args)) func(ref, args...);        // the trailing mask decides
        } else {
            func(ref, args..., true);
        }
    }
}

/// The i-th slice of a (nested) dynamic array as scalars / static arrays of scalars: slice(Array<HIPArray<float>, 3>, i) is the
/// Array<float, 3> (x_i, y_i, z_i) (array_struct.h:179-236, read-only here: entries of device arrays are copies).
template <typename T> inline auto slice(const T &a, size_t i) {
    if constexpr (!is_array_v<T>) {
        (void) i;
        return a;
    } else if constexpr (is_dynamic_array_v<T>) {
        if (i >= a.size() && a.size() != 1) throw std::out_of_range("slice(): index out of range");
        return a.coeff(a.size() == 1 ? 0 : i);               // size-1 arrays broadcast
    } else {
        Array<decltype(slice(a.coeff(0), i)), std::decay_t<T>::Size> r;
        for (size_t k = 0; k < std::decay_t<T>::Size; ++k) r.coeff(k) = slice(a.coeff(k), i);
        return r;
    }
}

/// the single entry of a size-1 array, or the scalar itself (array_router.h:1297-1307)
template <typename T> inline scalar_t<T> scalar_cast(const T &v) {
    static_assert(array_depth_v<T> <= 1, "scalar_cast(): scalars and flat arrays only");
    if constexpr (is_array_v<T>) {
        if (v.size() != 1) throw std::runtime_error("scalar_cast(): array should be of size 1!");
        return v.coeff(0);
    } else {
        return v;
    }
}

namespace detail {
    template <typename F> struct first_argument_of { };
    template <typename C, typename R, typename A> struct first_argument_of<R (C::*)(A) const> { using type = std::decay_t<A>; };
    template <typename C, typename R, typename A> struct first_argument_of<R (C::*)(A)> { using type = std::decay_t<A>; };
    template <typename F, typename = void> struct first_argument { };                    // generic lambdas, arrays: no member `type`
    template <typename F> struct first_argument<F, std::void_t<decltype(&F::operator())>> : first_argument_of<decltype(&F::operator())> { };
    template <typename R, typename A> struct first_argument<R (*)(A), void> { using type = std::decay_t<A>; };
}

/// Vectorised binary search (array_utils.h:130-171): the first index in [start, end) for which `pred(index)` is false,
/// assuming the predicate is true on a prefix.  Every lane searches its own answer: `pred` maps an index ARRAY to a mask;
/// the index type is taken from the predicate's parameter (or given explicitly for generic lambdas).
template <typename Index, typename Predicate>
inline Index binary_search(scalar_t<Index> start_, scalar_t<Index> end_, const Predicate &pred) {
    Index start(start_), end(end_);
    const size_t iterations = start_ < end_ ? (size_t) log2i((scalar_t<Index>) (end_ - start_)) + 1 : 0;
    for (size_t i = 0; i < iterations; ++i) {
        Index middle = sr<1>(start + end);
        mask_t<Index> cond = pred(middle);
        start = select(cond, min(middle + Index(scalar_t<Index>(1)), end), start);
        end = select(cond, end, middle);
    }
    return start;
}
template <typename Predicate, typename Index = typename detail::first_argument<Predicate>::type>
inline Index binary_search(scalar_t<Index> start_, scalar_t<Index> end_, const Predicate &pred) {
    return binary_search<Index, Predicate>(start_, end_, pred);
}

/// Angle between two unit vectors / between a unit vector and the z axis, well behaved near 0 and pi (array_math.h:1404-1436)
template <typename T> inline auto unit_angle(const T &a, const T &b) {
    auto dot_uv = dot(a, b);
    auto temp = 2.f * asin(.5f * norm(b - T(mulsign(value_t<T>(1.f), dot_uv)) * a));
    using E = decltype(temp);
    return select(dot_uv >= 0.f, temp, E(scalar_t<E>(3.14159265358979323846)) - temp);
}
template <typename T> inline auto unit_angle_z(const T &v) {
    static_assert(std::decay_t<T>::Size == 3, "unit_angle_z(): input is not a 3D vector");
    using E = value_t<T>;
    E temp = 2.f * asin(.5f * sqrt(sqr(v.x()) + sqr(v.y()) + sqr(v.z() - copysign(E(1.f), v.z()))));
    return select(v.z() >= 0.f, temp, E(scalar_t<E>(3.14159265358979323846)) - temp);
}

/// Neighbouring floating point values (array_math.h:1445-1495) and the denormal test (1497-1500)
template <typename T> inline T prev_float(const T &value) {
    using Int = int_array_t<T>;
    using IS = scalar_t<Int>;
    const Int exponent_mask(sizeof(IS) == 4 ? IS(0x7f800000) : IS(0x7ff0000000000000ll));
    const Int pos_denorm(sizeof(IS) == 4 ? IS(0x80000001) : IS(0x8000000000000001ll));
    Int i = reinterpret_array<Int>(value);
    auto is_nan_inf = eq(i & exponent_mask, exponent_mask), is_pos_0 = eq(i, Int(IS(0))), is_gt_0 = i >= Int(IS(0));
    Int j1 = i + select(is_gt_0, Int(IS(-1)), Int(IS(1))), j2 = select(is_pos_0, pos_denorm, i);
    return reinterpret_array<T>(select(is_nan_inf | is_pos_0, j2, j1));
}
template <typename T> inline T next_float(const T &value) {
    using Int = int_array_t<T>;
    using IS = scalar_t<Int>;
    const Int exponent_mask(sizeof(IS) == 4 ? IS(0x7f800000) : IS(0x7ff0000000000000ll));
    const Int sign_mask(sizeof(IS) == 4 ? IS(0x80000000) : IS(0x8000000000000000ll));
    Int i = reinterpret_array<Int>(value);
    auto is_nan_inf = eq(i & exponent_mask, exponent_mask), is_neg_0 = eq(i, sign_mask), is_gt_0 = i >= Int(IS(0));
    Int j1 = i + select(is_gt_0, Int(IS(1)), Int(IS(-1))), j2 = select(is_neg_0, Int(IS(1)), i);
    return reinterpret_array<T>(select(is_nan_inf | is_neg_0, j2, j1));
}
template <typename T> inline auto isdenormal(const T &a) {
    return (abs(a) < T(std::numeric_limits<scalar_t<T>>::min())) & neq(a, T(scalar_t<T>(0)));
}
template <typename T1, typename T2> inline auto copysign_neg(const T1 &a, const T2 &b) { return copysign(a, -b); }
template <typename T1, typename T2> inline auto mulsign_neg(const T1 &a, const T2 &b) { return mulsign(a, -b); }
template <typename... Args> inline void prefetch(const Args &...) { }       // no counterpart on the device

/// even components a * b - c, odd components a * b + c (and the other way round)
template <typename V, size_t N> inline Array<V, N> fmaddsub(const Array<V, N> &a, const Array<V, N> &b, const Array<V, N> &c) {
    Array<V, N> r;
    for (size_t i = 0; i < N; ++i) r.coeff(i) = (i % 2 == 0) ? fmsub(a.coeff(i), b.coeff(i), c.coeff(i)) : fmadd(a.coeff(i), b.coeff(i), c.coeff(i));
    return r;
}
template <typename V, size_t N> inline Array<V, N> fmsubadd(const Array<V, N> &a, const Array<V, N> &b, const Array<V, N> &c) {
    Array<V, N> r;
    for (size_t i = 0; i < N; ++i) r.coeff(i) = (i % 2 == 0) ? fmadd(a.coeff(i), b.coeff(i), c.coeff(i)) : fmsub(a.coeff(i), b.coeff(i), c.coeff(i));
    return r;
}
/// rotate the COMPONENTS of a static array: rol_array<1>({a, b, c}) = {b, c, a}
template <size_t Imm, typename V, size_t N> inline Array<V, N> rol_array(const Array<V, N> &a) {
    Array<V, N> r;
    for (size_t i = 0; i < N; ++i) r.coeff(i) = a.coeff((i + Imm) % N);
    return r;
}
template <size_t Imm, typename V, size_t N> inline Array<V, N> ror_array(const Array<V, N> &a) {
    Array<V, N> r;
    for (size_t i = 0; i < N; ++i) r.coeff(i) = a.coeff((i + N - Imm % N) % N);
    return r;
}
namespace detail {
    constexpr size_t fill_bits(size_t i) { return i != 0 ? i | fill_bits(i >> 1) : 0; }
    constexpr size_t lpow2(size_t i) { return i != 0 ? (fill_bits(i - 1) >> 1) + 1 : 0; }      // largest power of two below i
}
/// the two parts a static array splits into: the low part has the largest power-of-two size below N
template <typename V, size_t N> inline auto low(const Array<V, N> &a) { return head<detail::lpow2(N)>(a); }
template <typename V, size_t N> inline auto high(const Array<V, N> &a) { return tail<N - detail::lpow2(N)>(a); }

/// a == b / a != b reduce to a single bool: all (resp. any) entries compare equal (unequal), array_router.h:494-503.
/// (Entry-wise comparisons are eq() / neq().)
template <typename T1, typename T2, enable_if_array_any_t<T1, T2> = 0> inline bool operator==(const T1 &a, const T2 &b) {
    return all_nested(eq(a, b));
}
template <typename T1, typename T2, enable_if_array_any_t<T1, T2> = 0> inline bool operator!=(const T1 &a, const T2 &b) {
    return any_nested(neq(a, b));
}

/// |a - b| <= |b| rtol + atol for every entry (array_router.h:1309-1321)
template <typename T1, typename T2> inline bool allclose(const T1 &a, const T2 &b, float rtol = 1e-5f, float atol = 1e-8f,
                                                         bool equal_nan = false) {
    if constexpr (!is_array_v<T1> && !is_array_v<T2>) {
        using S = decltype(a - b);
        S d = a - b, lim = (b < 0 ? -b : b) * S(rtol) + S(atol);
        return ((d < 0 ? -d : d) <= lim) || (equal_nan && a != a && b != b);
    } else {
        using E = expr_t<T1, T2>;
        using S = scalar_t<E>;
        const E ea = detail::as<E>(a), eb = detail::as<E>(b);
        auto cond = abs(ea - eb) <= abs(eb) * E(S(rtol)) + E(S(atol));
        if constexpr (std::is_floating_point_v<S>) {
            if (equal_nan) cond = cond | (isnan(ea) & isnan(eb));
        }
        return all_nested(cond);
    }
}

// ---------------------------------------------------------------------------------------------
//  Masked assignment: masked(x, m) = v;  masked(x, m) += v;  x[m] = v   (array_masked.h, array_base.h:144-157 --
//  on dynamic arrays every variant is a select)
// ---------------------------------------------------------------------------------------------
namespace detail {
    template <typename T> struct MaskedArray {
        using Mask = std::conditional_t<is_array_v<T> || is_struct_v<T>, mask_t<std::conditional_t<is_struct_v<T>, bool, T>>, bool>;
        T &d;
        Mask m;
        template <typename V> void operator=(const V &v) { assign(T(v)); }
        template <typename V> void operator+=(const V &v) { assign(d + T(v)); }
        template <typename V> void operator-=(const V &v) { assign(d - T(v)); }
        template <typename V> void operator*=(const V &v) { assign(d * T(v)); }
        template <typename V> void operator/=(const V &v) { assign(d / T(v)); }
        template <typename V> void operator|=(const V &v) { assign(d | T(v)); }
        template <typename V> void operator&=(const V &v) { assign(d & T(v)); }
        template <typename V> void operator^=(const V &v) { assign(d ^ T(v)); }
    private:
        void assign(const T &v) {
            if constexpr (is_array_v<T> || is_struct_v<T>) d = select(m, v, d);
            else if (m) d = v;
        }
    };
    template <typename T, typename M> struct MaskedStruct {      // ENOKI_STRUCT types: field-wise select with any mask type
        T &d;
        M m;
        void operator=(const T &v) { d = select(m, v, d); }
    };
}

template <typename T, typename M, enable_if_t<!is_struct_v<T>> = 0> inline detail::MaskedArray<T> masked(T &value, const M &mask) {
    return detail::MaskedArray<T>{ value, typename detail::MaskedArray<T>::Mask(mask) };
}
template <typename T, typename M, enable_if_t<is_struct_v<T>> = 0> inline detail::MaskedStruct<T, M> masked(T &value, const M &mask) {
    return detail::MaskedStruct<T, M>{ value, mask };
}

template <typename Value_, size_t Size_> template <typename M, enable_if_t<is_mask_v<M>>>
inline auto Array<Value_, Size_>::operator[](const M &mask) { return masked(*this, mask); }

// ---------------------------------------------------------------------------------------------
//  Autodiff helpers that are no-ops for non-differentiable types (autodiff.h:1414-1500)
// ---------------------------------------------------------------------------------------------

template <typename T> inline decltype(auto) detach(const T &a) {
    if constexpr (!is_diff_array_v<T>) {
        return (const T &) a;
    } else if constexpr (std::decay_t<T>::Depth > 1) {           // Array<DiffArray<...>, N>: component by component
        using V = std::decay_t<decltype(detach(a.coeff(0)))>;
        Array<V, std::decay_t<T>::Size> result;
        for (size_t i = 0; i < std::decay_t<T>::Size; ++i) result.coeff(i) = detach(a.coeff(i));
        return result;
    } else {
        return a.value_();
    }
}

} // namespace enoki
