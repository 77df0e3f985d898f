// The unary operations of the elementwise kernels as device functors, shared with the kernels that apply one while
// loading their input: horizontal reductions over a mapped array (reduce.hip, ek_hip_reduce_map) and the value streams of
// scatter_add (scatter_binned.hip, ek_hip_scatter_add_multi_map).  One definition = one rounding behaviour everywhere.
#pragma once
#include "ek_map.h"
#include <enoki/device/ek_math.h>
#include <enoki/device/ek_special.h>

namespace ek {

template <typename T> inline constexpr bool is_fp = std::is_floating_point_v<T>;
template <typename T> inline constexpr bool is_int = std::is_integral_v<T> && !std::is_same_v<T, uint8_t>;
template <typename T> inline constexpr bool is_mask = std::is_same_v<T, uint8_t>;

template <typename T> using uint_of = std::conditional_t<sizeof(T) == 8, uint64_t, std::conditional_t<sizeof(T) == 4, uint32_t, uint8_t>>;

template <typename T> __device__ __forceinline__ uint_of<T> bits(T v) {
    uint_of<T> u;
    __builtin_memcpy(&u, &v, sizeof(T));
    return u;
}
template <typename T> __device__ __forceinline__ T from_bits(uint_of<T> u) {
    T v;
    __builtin_memcpy(&v, &u, sizeof(T));
    return v;
}

// ------------------------------------------------------------------------------------------------
//  Unary
// ------------------------------------------------------------------------------------------------
template <int Op, typename T> constexpr bool unary_supported() {
    switch (Op) {
        case EK_NEG: case EK_ABS: return !is_mask<T>;
        case EK_NOT: return !is_fp<T>;
        case EK_SQRT: case EK_RCP: case EK_RSQRT: case EK_FLOOR: case EK_CEIL: case EK_ROUND: case EK_TRUNC:
        case EK_SIGN: case EK_RCP_SQR: case EK_RSQRT_SQR: case EK_RSQRT_CUBE: case EK_SEC_SQR: case EK_SECH_SQR: case EK_RCP_1P_SQR: return is_fp<T>;
        case EK_SIN: case EK_COS: case EK_EXP: case EK_LOG:
        case EK_TAN: case EK_COT: case EK_ASIN: case EK_ACOS: case EK_ATAN: case EK_SINH: case EK_COSH: case EK_TANH:
        case EK_ASINH: case EK_ACOSH: case EK_ATANH: case EK_CBRT: return is_fp<T>;
        case EK_ERF: case EK_ERFC: case EK_ERFINV: case EK_I0E: case EK_DAWSON: case EK_ERFI: case EK_LGAMMA: case EK_TGAMMA:
            return is_fp<T>;
        case EK_POPCNT: case EK_LZCNT: case EK_TZCNT: return is_int<T>;
        case EK_COPY: return true;
        default: return false;
    }
}

template <int Op, typename T> struct UnaryOp {
    static __device__ __forceinline__ T apply(T x) {
        using U = uint_of<T>;
        constexpr U sign_bit = U(1) << (sizeof(T) * 8 - 1);
        if constexpr (Op == EK_COPY) {
            return x;
        } else if constexpr (Op == EK_NEG) {
            if constexpr (is_fp<T>) return from_bits<T>(bits(x) ^ sign_bit);
            else return (T) (U(0) - (U) x);
        } else if constexpr (Op == EK_ABS) {
            if constexpr (is_fp<T>) return from_bits<T>(bits(x) & ~sign_bit);
            else if constexpr (std::is_signed_v<T>) return x < 0 ? (T) (U(0) - (U) x) : x;
            else return x;
        } else if constexpr (Op == EK_NOT) {
            if constexpr (is_mask<T>) return x ? 0 : 1;
            else return (T) ~(U) x;
        } else if constexpr (Op == EK_SQRT) {
            if constexpr (sizeof(T) == 4) return __builtin_sqrtf(x); else return __builtin_sqrt(x);   // correctly rounded expansion
        } else if constexpr (Op == EK_RCP) {
            return T(1) / x;
        } else if constexpr (Op == EK_RSQRT) {
            if constexpr (sizeof(T) == 4) return 1.0f / __builtin_sqrtf(x); else return 1.0 / __builtin_sqrt(x);
        } else if constexpr (Op == EK_RCP_SQR) {
            const T r = T(1) / x;
            return r * r;
        } else if constexpr (Op == EK_SEC_SQR) {
            const T r = T(1) / UnaryOp<EK_COS, T>::apply(x);         // sqr(sec(x)), sec = rcp(cos): d/dx tan (autodiff.h:532-541)
            return r * r;
        } else if constexpr (Op == EK_SECH_SQR) {
            const T r = T(1) / UnaryOp<EK_COSH, T>::apply(x);        // sqr(sech(x)), sech = rcp(cosh): d/dx tanh (autodiff.h:685-696)
            return r * r;
        } else if constexpr (Op == EK_RCP_1P_SQR) {
            return T(1) / (T(1) + x * x);                             // rcp(1 + sqr(x)): d/dx atan (autodiff.h:606-616)
        } else if constexpr (Op == EK_RSQRT_SQR || Op == EK_RSQRT_CUBE) {
            T r;
            if constexpr (sizeof(T) == 4) r = 1.0f / __builtin_sqrtf(x); else r = 1.0 / __builtin_sqrt(x);
            const T r2 = r * r;
            if constexpr (Op == EK_RSQRT_SQR) return r2; else return r * r2;
        } else if constexpr (Op == EK_FLOOR) {
            if constexpr (sizeof(T) == 4) return __builtin_floorf(x); else return __builtin_floor(x);
        } else if constexpr (Op == EK_CEIL) {
            if constexpr (sizeof(T) == 4) return __builtin_ceilf(x); else return __builtin_ceil(x);
        } else if constexpr (Op == EK_ROUND) {
            if constexpr (sizeof(T) == 4) return __builtin_rintf(x); else return __builtin_rint(x);
        } else if constexpr (Op == EK_TRUNC) {
            if constexpr (sizeof(T) == 4) return __builtin_truncf(x); else return __builtin_trunc(x);
        } else if constexpr (Op == EK_SIGN) {
            // (sign_mask & a) | 1.0   (array_router.h:371)
            return from_bits<T>((bits(x) & sign_bit) | bits(T(1)));
        } else if constexpr (Op == EK_SIN) {
            T s, c;
            if constexpr (sizeof(T) == 4) dev::sincos_f32<true, false>(x, s, c); else dev::sincos_f64<true, false>(x, s, c);
            return s;
        } else if constexpr (Op == EK_COS) {
            T s, c;
            if constexpr (sizeof(T) == 4) dev::sincos_f32<false, true>(x, s, c); else dev::sincos_f64<false, true>(x, s, c);
            return c;
        } else if constexpr (Op == EK_EXP) {
            if constexpr (sizeof(T) == 4) return dev::exp_f32(x); else return dev::exp_f64(x);
        } else if constexpr (Op == EK_LOG) {
            if constexpr (sizeof(T) == 4) return dev::log_f32(x); else return dev::log_f64(x);
        } else if constexpr (Op == EK_TAN) {
            if constexpr (sizeof(T) == 4) return dev::tancot_f32<true>(x); else return dev::tancot_f64<true>(x);
        } else if constexpr (Op == EK_COT) {
            if constexpr (sizeof(T) == 4) return dev::tancot_f32<false>(x); else return dev::tancot_f64<false>(x);
        } else if constexpr (Op == EK_ASIN) {
            if constexpr (sizeof(T) == 4) return dev::asin_f32(x); else return dev::asin_f64(x);
        } else if constexpr (Op == EK_ACOS) {
            if constexpr (sizeof(T) == 4) return dev::acos_f32(x); else return dev::acos_f64(x);
        } else if constexpr (Op == EK_ATAN) {
            if constexpr (sizeof(T) == 4) return dev::atan2_f32(x, 1.0f); else return dev::atan2_f64(x, 1.0);   // array_math.h:666-668
        } else if constexpr (Op == EK_SINH) {
            if constexpr (sizeof(T) == 4) return dev::sinh_f32(x); else return dev::sinh_f64(x);
        } else if constexpr (Op == EK_COSH) {
            if constexpr (sizeof(T) == 4) return dev::cosh_f32(x); else return dev::cosh_f64(x);
        } else if constexpr (Op == EK_TANH) {
            if constexpr (sizeof(T) == 4) return dev::tanh_f32(x); else return dev::tanh_f64(x);
        } else if constexpr (Op == EK_ASINH) {
            if constexpr (sizeof(T) == 4) return dev::asinh_f32(x); else return dev::asinh_f64(x);
        } else if constexpr (Op == EK_ACOSH) {
            if constexpr (sizeof(T) == 4) return dev::acosh_f32(x); else return dev::acosh_f64(x);
        } else if constexpr (Op == EK_ATANH) {
            if constexpr (sizeof(T) == 4) return dev::atanh_f32(x); else return dev::atanh_f64(x);
        } else if constexpr (Op == EK_CBRT) {
            if constexpr (sizeof(T) == 4) return dev::cbrt_f32(x); else return dev::cbrt_f64(x);
        } else if constexpr (Op == EK_ERF) {
            if constexpr (is_fp<T>) return dev::erf_t<T>(x); else return x;
        } else if constexpr (Op == EK_ERFC) {
            if constexpr (is_fp<T>) return dev::erfc_t<T>(x); else return x;
        } else if constexpr (Op == EK_ERFINV) {
            if constexpr (is_fp<T>) return dev::erfinv_t<T>(x); else return x;
        } else if constexpr (Op == EK_I0E) {
            if constexpr (is_fp<T>) return dev::i0e_t<T>(x); else return x;
        } else if constexpr (Op == EK_DAWSON) {
            if constexpr (is_fp<T>) return dev::dawson_t<T>(x); else return x;
        } else if constexpr (Op == EK_ERFI) {
            if constexpr (is_fp<T>) return dev::erfi_t<T>(x); else return x;
        } else if constexpr (Op == EK_LGAMMA) {
            if constexpr (is_fp<T>) return dev::lgamma_t<T>(x); else return x;
        } else if constexpr (Op == EK_TGAMMA) {
            if constexpr (is_fp<T>) return dev::tgamma_t<T>(x); else return x;
        } else if constexpr (Op == EK_POPCNT) {
            if constexpr (sizeof(T) == 4) return (T) __popc((uint32_t) x); else return (T) __popcll((uint64_t) x);
        } else if constexpr (Op == EK_LZCNT) {
            if constexpr (sizeof(T) == 4) return (T) (x ? __clz((int) x) : 32); else return (T) (x ? __clzll((long long) x) : 64);
        } else if constexpr (Op == EK_TZCNT) {
            if constexpr (sizeof(T) == 4) return (T) (x ? __ffs((int) x) - 1 : 32); else return (T) (x ? __ffsll((long long) x) - 1 : 64);
        } else {
            return x;
        }
    }
};

struct SinCosOp {
    static __device__ __forceinline__ void apply(float x, float &s, float &c) { dev::sincos_f32<true, true>(x, s, c); }
    static __device__ __forceinline__ void apply(double x, double &s, double &c) { dev::sincos_f64<true, true>(x, s, c); }
};

// runtime-selected fusable op (wave-uniform `op`); EK_COPY = identity
template <typename T> __device__ __forceinline__ T unary_fused(int op, T x) {
    switch (op) {
        case EK_NEG: return UnaryOp<EK_NEG, T>::apply(x);
        case EK_ABS: return UnaryOp<EK_ABS, T>::apply(x);
        case EK_SQRT: return UnaryOp<EK_SQRT, T>::apply(x);
        case EK_RCP: return UnaryOp<EK_RCP, T>::apply(x);
        case EK_RSQRT: return UnaryOp<EK_RSQRT, T>::apply(x);
        case EK_SIN: return UnaryOp<EK_SIN, T>::apply(x);
        case EK_COS: return UnaryOp<EK_COS, T>::apply(x);
        case EK_EXP: return UnaryOp<EK_EXP, T>::apply(x);
        case EK_LOG: return UnaryOp<EK_LOG, T>::apply(x);
        case EK_RCP_SQR: return UnaryOp<EK_RCP_SQR, T>::apply(x);
        case EK_RSQRT_SQR: return UnaryOp<EK_RSQRT_SQR, T>::apply(x);
        case EK_RSQRT_CUBE: return UnaryOp<EK_RSQRT_CUBE, T>::apply(x);
        default: return x;      // (the second-wave maps of unary_chainable() are applied by the chain kernels only: reduce.hip)
    }
}

} // namespace ek
