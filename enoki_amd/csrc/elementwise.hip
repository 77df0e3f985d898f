// Vertical (elementwise) ops of HIPArray<T>: arithmetic, fma family, rounding, bit ops,
// transcendental first wave, compares, select, casts, and the fused backward primitives
// safe_mul / safe_fmadd.  SURVEY.md rows a2-a5, a11.
//
// Semantics follow the reference's CPU (AVX2 DynamicArray) path, which is the parity target:
//   * IEEE round-to-nearest add/sub/mul/div/sqrt/fma, no flush-to-zero (the CUDA path's .ftz,
//     cuda.h:343-430, is NOT reproduced: the CPU path does not flush, array_intrin.h:167-194);
//   * min/max: first operand wins on unordered compares (array_avx.h:244-245);
//   * f32 -> i32 casts truncate, out of range gives 0x80000000 (cvttps2dq);
//   * variable shifts with count >= width give 0 / sign fill (vpsllvd & co., array_avx2.h);
//   * sin/cos/exp/log: array_math.h algorithms (include/enoki/device/ek_math.h), bit-exact.
#include "ek_unary.h"

namespace ek {

// kernel names reported by ek_hip_profile_end() / ENOKI_HIP_LOG=3
static const char *const unary_names[EK_UNARY_COUNT] = {
    "neg", "abs", "not", "sqrt", "rcp", "rsqrt", "floor", "ceil", "round", "trunc", "sin", "cos", "exp", "log",
    "popcnt", "lzcnt", "tzcnt", "sign", "copy", "tan", "cot", "asin", "acos", "atan", "sinh", "cosh", "tanh", "asinh",
    "acosh", "atanh", "cbrt", "erf", "erfc", "erfinv", "i0e", "dawson", "erfi", "lgamma", "tgamma", "rcp_sqr", "rsqrt_sqr", "rsqrt_cube",
    "sec_sqr", "sech_sqr", "rcp_1p_sqr" };
static const char *const binary_names[EK_BINARY_COUNT] = {
    "add", "sub", "mul", "div", "mod", "min", "max", "mulhi", "and", "or", "xor", "sl", "sr", "safe_mul",
    "atan2", "pow", "fmod", "ldexp" };
static const char *const ternary_names[EK_TERNARY_COUNT] = { "fmadd", "fmsub", "fnmadd", "fnmsub", "safe_fmadd", "muladd", "mulsub", "nmuladd" };


struct SinCoshOp {
    static __device__ __forceinline__ void apply(float x, float &s, float &c) { dev::sincosh_f32(x, s, c); }
    static __device__ __forceinline__ void apply(double x, double &s, double &c) { dev::sincosh_f64(x, s, c); }
};

template <int Op, typename T> int unary_launch(void *out, const ek_operand *a, size_t n) {
    if constexpr (unary_supported<Op, T>()) {
        Arg<T> aa;
        if (int rc = make_arg<T>(a, n, aa, "ek_hip_unary")) return rc;
        return launch_map1<UnaryOp<Op, T>>(unary_names[Op], (T *) out, n, aa);
    } else {
        return fail(EK_ERR_UNSUPPORTED, "ek_hip_unary(): op %d is not defined for type %d", Op, (int) sizeof(T));
    }
}

#define EK_UNARY_CASE(OP) case OP: return unary_launch<OP, T>(out, a, n);
template <typename T> int unary_dispatch(int op, void *out, const ek_operand *a, size_t n) {
    switch (op) {
        EK_UNARY_CASE(EK_NEG) EK_UNARY_CASE(EK_ABS) EK_UNARY_CASE(EK_NOT) EK_UNARY_CASE(EK_SQRT)
        EK_UNARY_CASE(EK_RCP) EK_UNARY_CASE(EK_RSQRT) EK_UNARY_CASE(EK_FLOOR) EK_UNARY_CASE(EK_CEIL)
        EK_UNARY_CASE(EK_ROUND) EK_UNARY_CASE(EK_TRUNC) EK_UNARY_CASE(EK_SIN) EK_UNARY_CASE(EK_COS)
        EK_UNARY_CASE(EK_EXP) EK_UNARY_CASE(EK_LOG) EK_UNARY_CASE(EK_POPCNT) EK_UNARY_CASE(EK_LZCNT)
        EK_UNARY_CASE(EK_TZCNT) EK_UNARY_CASE(EK_SIGN) EK_UNARY_CASE(EK_COPY)
        EK_UNARY_CASE(EK_TAN) EK_UNARY_CASE(EK_COT) EK_UNARY_CASE(EK_ASIN) EK_UNARY_CASE(EK_ACOS)
        EK_UNARY_CASE(EK_ATAN) EK_UNARY_CASE(EK_SINH) EK_UNARY_CASE(EK_COSH) EK_UNARY_CASE(EK_TANH)
        EK_UNARY_CASE(EK_ASINH) EK_UNARY_CASE(EK_ACOSH) EK_UNARY_CASE(EK_ATANH) EK_UNARY_CASE(EK_CBRT)
        EK_UNARY_CASE(EK_ERF) EK_UNARY_CASE(EK_ERFC) EK_UNARY_CASE(EK_ERFINV) EK_UNARY_CASE(EK_I0E) EK_UNARY_CASE(EK_DAWSON)
        EK_UNARY_CASE(EK_ERFI) EK_UNARY_CASE(EK_LGAMMA) EK_UNARY_CASE(EK_TGAMMA)
        EK_UNARY_CASE(EK_RCP_SQR) EK_UNARY_CASE(EK_RSQRT_SQR) EK_UNARY_CASE(EK_RSQRT_CUBE)
        EK_UNARY_CASE(EK_SEC_SQR) EK_UNARY_CASE(EK_SECH_SQR) EK_UNARY_CASE(EK_RCP_1P_SQR)
        default: return fail(EK_ERR_INVALID, "ek_hip_unary(): unknown op %d", op);
    }
}

// ------------------------------------------------------------------------------------------------
//  Binary
// ------------------------------------------------------------------------------------------------
template <int Op, typename T> constexpr bool binary_supported() {
    switch (Op) {
        case EK_ADD: case EK_SUB: case EK_MUL: case EK_DIV: case EK_MIN: case EK_MAX: return !is_mask<T>;
        case EK_MOD: case EK_MULHI: case EK_SL: case EK_SR: return is_int<T>;
        case EK_AND: case EK_OR: case EK_XOR: return true;   // fp: bitwise on the representation
        case EK_SAFE_MUL: case EK_FMOD: return is_fp<T>;
        case EK_ATAN2: case EK_POW: case EK_LDEXP: return is_fp<T>;
        default: return false;
    }
}

template <int Op, typename T> struct BinaryOp {
    static __device__ __forceinline__ T apply(T x, T y) {
        using U = uint_of<T>;
        constexpr unsigned Bits = sizeof(T) * 8;
        if constexpr (Op == EK_ADD) {
            if constexpr (is_fp<T>) return x + y; else return (T) ((U) x + (U) y);
        } else if constexpr (Op == EK_SUB) {
            if constexpr (is_fp<T>) return x - y; else return (T) ((U) x - (U) y);
        } else if constexpr (Op == EK_MUL) {
            if constexpr (is_fp<T>) return x * y; else return (T) ((U) x * (U) y);
        } else if constexpr (Op == EK_DIV) {
            if constexpr (is_fp<T>) return x / y;
            else return y == 0 ? T(0) : ((std::is_signed_v<T> && y == T(-1)) ? (T) (U(0) - (U) x) : (T) (x / y));
        } else if constexpr (Op == EK_MOD) {
            return y == 0 ? T(0) : ((std::is_signed_v<T> && y == T(-1)) ? T(0) : (T) (x % y));
        } else if constexpr (Op == EK_MIN) {
            return y < x ? y : x;
        } else if constexpr (Op == EK_MAX) {
            return y > x ? y : x;
        } else if constexpr (Op == EK_MULHI) {
            if constexpr (std::is_same_v<T, int32_t>) return __mulhi(x, y);
            else if constexpr (std::is_same_v<T, uint32_t>) return __umulhi(x, y);
            else if constexpr (std::is_same_v<T, int64_t>) return __mul64hi(x, y);
            else return __umul64hi(x, y);
        } else if constexpr (Op == EK_AND) {
            return from_bits<T>(bits(x) & bits(y));
        } else if constexpr (Op == EK_OR) {
            return from_bits<T>(bits(x) | bits(y));
        } else if constexpr (Op == EK_XOR) {
            return from_bits<T>(bits(x) ^ bits(y));
        } else if constexpr (Op == EK_SL) {
            return (U) y >= Bits ? T(0) : (T) ((U) x << (U) y);
        } else if constexpr (Op == EK_SR) {
            if constexpr (std::is_signed_v<T>) return (U) y >= Bits ? (x < 0 ? T(-1) : T(0)) : (T) (x >> (U) y);
            else return (U) y >= Bits ? T(0) : (T) (x >> y);
        } else if constexpr (Op == EK_SAFE_MUL) {
            return dev::safe_mul(x, y);
        } else if constexpr (Op == EK_ATAN2) {
            if constexpr (sizeof(T) == 4) return dev::atan2_f32(x, y); else return dev::atan2_f64(x, y);
        } else if constexpr (Op == EK_POW) {
            if constexpr (sizeof(T) == 4) return dev::pow_f32(x, y); else return dev::pow_f64(x, y);
        } else if constexpr (Op == EK_LDEXP) {
            if constexpr (sizeof(T) == 4) return dev::ldexp_f32(x, y); else return dev::ldexp_f64(x, y);
        } else if constexpr (Op == EK_FMOD) {
            if constexpr (sizeof(T) == 4) return dev::fmod_f32(x, y);
            else return __builtin_fma(-__builtin_trunc(x / y), y, x);
        } else {
            return x;
        }
    }
};

template <int Op, typename T> int binary_launch(void *out, const ek_operand *a, const ek_operand *b, size_t n) {
    if constexpr (binary_supported<Op, T>()) {
        Arg<T> aa, bb;
        if (int rc = make_arg<T>(a, n, aa, "ek_hip_binary")) return rc;
        if (int rc = make_arg<T>(b, n, bb, "ek_hip_binary")) return rc;
        return launch_map2<BinaryOp<Op, T>>(binary_names[Op], (T *) out, n, aa, bb);
    } else {
        return fail(EK_ERR_UNSUPPORTED, "ek_hip_binary(): op %d is not defined for this type", Op);
    }
}

#define EK_BINARY_CASE(OP) case OP: return binary_launch<OP, T>(out, a, b, n);
template <typename T> int binary_dispatch(int op, void *out, const ek_operand *a, const ek_operand *b, size_t n) {
    switch (op) {
        EK_BINARY_CASE(EK_ADD) EK_BINARY_CASE(EK_SUB) EK_BINARY_CASE(EK_MUL) EK_BINARY_CASE(EK_DIV)
        EK_BINARY_CASE(EK_MOD) EK_BINARY_CASE(EK_MIN) EK_BINARY_CASE(EK_MAX) EK_BINARY_CASE(EK_MULHI)
        EK_BINARY_CASE(EK_AND) EK_BINARY_CASE(EK_OR) EK_BINARY_CASE(EK_XOR) EK_BINARY_CASE(EK_SL)
        EK_BINARY_CASE(EK_SR) EK_BINARY_CASE(EK_SAFE_MUL) EK_BINARY_CASE(EK_ATAN2) EK_BINARY_CASE(EK_POW)
        EK_BINARY_CASE(EK_FMOD) EK_BINARY_CASE(EK_LDEXP)
        default: return fail(EK_ERR_INVALID, "ek_hip_binary(): unknown op %d", op);
    }
}

// ------------------------------------------------------------------------------------------------
//  Ternary
// ------------------------------------------------------------------------------------------------
template <int Op, typename T> struct TernaryOp {
    static __device__ __forceinline__ T apply(T x, T y, T z) {
        using U = uint_of<T>;
        if constexpr (is_fp<T>) {
            auto fma_ = [](T a, T b, T c) -> T {
                if constexpr (sizeof(T) == 4) return __builtin_fmaf(a, b, c); else return __builtin_fma(a, b, c);
            };
            if constexpr (Op == EK_FMADD) return fma_(x, y, z);
            else if constexpr (Op == EK_FMSUB) return fma_(x, y, -z);
            else if constexpr (Op == EK_FNMADD) return fma_(-x, y, z);
            else if constexpr (Op == EK_FNMSUB) return fma_(-x, y, -z);
            // two roundings (the translation unit is built with -ffp-contract=off: a product and a sum never fuse)
            else if constexpr (Op == EK_MULADD) return x * y + z;
            else if constexpr (Op == EK_MULSUB) return x * y - z;
            else if constexpr (Op == EK_NMULADD) return z - x * y;
            else return dev::safe_fmadd(x, y, z);
        } else {
            // integer mad.lo (cuda.h:387-394)
            U p = (U) x * (U) y;
            if constexpr (Op == EK_FMADD) return (T) (p + (U) z);
            else if constexpr (Op == EK_FMSUB) return (T) (p - (U) z);
            else if constexpr (Op == EK_FNMADD) return (T) ((U) z - p);
            else return (T) (U(0) - p - (U) z);
        }
    }
};

template <int Op, typename T>
int ternary_launch(void *out, const ek_operand *a, const ek_operand *b, const ek_operand *c, size_t n) {
    if constexpr (is_mask<T> || ((Op == EK_SAFE_FMADD || Op == EK_MULADD || Op == EK_MULSUB || Op == EK_NMULADD) && !is_fp<T>)) {
        return fail(EK_ERR_UNSUPPORTED, "ek_hip_ternary(): op %d is not defined for this type", Op);
    } else {
        Arg<T> aa, bb, cc;
        if (int rc = make_arg<T>(a, n, aa, "ek_hip_ternary")) return rc;
        if (int rc = make_arg<T>(b, n, bb, "ek_hip_ternary")) return rc;
        if (int rc = make_arg<T>(c, n, cc, "ek_hip_ternary")) return rc;
        return launch_map3<TernaryOp<Op, T>>(ternary_names[Op], (T *) out, n, aa, bb, cc);
    }
}

#define EK_TERNARY_CASE(OP) case OP: return ternary_launch<OP, T>(out, a, b, c, n);
template <typename T>
int ternary_dispatch(int op, void *out, const ek_operand *a, const ek_operand *b, const ek_operand *c, size_t n) {
    switch (op) {
        EK_TERNARY_CASE(EK_FMADD) EK_TERNARY_CASE(EK_FMSUB) EK_TERNARY_CASE(EK_FNMADD)
        EK_TERNARY_CASE(EK_FNMSUB) EK_TERNARY_CASE(EK_SAFE_FMADD)
        EK_TERNARY_CASE(EK_MULADD) EK_TERNARY_CASE(EK_MULSUB) EK_TERNARY_CASE(EK_NMULADD)
        default: return fail(EK_ERR_INVALID, "ek_hip_ternary(): unknown op %d", op);
    }
}

// ------------------------------------------------------------------------------------------------
//  Compare / select / cast
// ------------------------------------------------------------------------------------------------
template <int Op, typename T> struct CompareOp {
    static __device__ __forceinline__ uint8_t apply(T x, T y) {
        if constexpr (Op == EK_EQ) return x == y;
        else if constexpr (Op == EK_NEQ) return x != y;
        else if constexpr (Op == EK_LT) return x < y;
        else if constexpr (Op == EK_LE) return x <= y;
        else if constexpr (Op == EK_GT) return x > y;
        else return x >= y;
    }
};

template <int Op, typename T> int compare_launch(uint8_t *out, const ek_operand *a, const ek_operand *b, size_t n) {
    Arg<T> aa, bb;
    if (int rc = make_arg<T>(a, n, aa, "ek_hip_compare")) return rc;
    if (int rc = make_arg<T>(b, n, bb, "ek_hip_compare")) return rc;
    return launch_map2<CompareOp<Op, T>>("compare", out, n, aa, bb);
}

template <typename T> int compare_dispatch(int op, uint8_t *out, const ek_operand *a, const ek_operand *b, size_t n) {
    switch (op) {
        case EK_EQ: return compare_launch<EK_EQ, T>(out, a, b, n);
        case EK_NEQ: return compare_launch<EK_NEQ, T>(out, a, b, n);
        case EK_LT: return compare_launch<EK_LT, T>(out, a, b, n);
        case EK_LE: return compare_launch<EK_LE, T>(out, a, b, n);
        case EK_GT: return compare_launch<EK_GT, T>(out, a, b, n);
        case EK_GE: return compare_launch<EK_GE, T>(out, a, b, n);
        default: return fail(EK_ERR_INVALID, "ek_hip_compare(): unknown op %d", op);
    }
}

template <typename T> struct SelectOp {
    static __device__ __forceinline__ T apply(uint8_t m, T t, T f) { return m ? t : f; }
};

template <typename T>
int select_launch(void *out, const ek_operand *m, const ek_operand *t, const ek_operand *f, size_t n) {
    Arg<uint8_t> mm;
    Arg<T> tt, ff;
    if (int rc = make_arg<uint8_t>(m, n, mm, "ek_hip_select")) return rc;
    if (int rc = make_arg<T>(t, n, tt, "ek_hip_select")) return rc;
    if (int rc = make_arg<T>(f, n, ff, "ek_hip_select")) return rc;
    // a mask vector is read 16/sizeof(T) bytes per lane: needs that alignment only
    return launch_map3<SelectOp<T>>("select", (T *) out, n, mm, tt, ff);
}

template <typename S, typename D> struct CastOp {
    static __device__ __forceinline__ D apply(S x) {
        if constexpr (std::is_same_v<S, float> && std::is_same_v<D, int32_t>) {
            return dev::cvtt_i32(x);
        } else if constexpr (is_fp<S> && std::is_integral_v<D>) {
            // truncation (cvt.rzi, cuda.h:239-240); clamp like the hardware converter
            if constexpr (std::is_same_v<D, uint8_t>) return x != S(0);
            else return (D) x;
        } else if constexpr (std::is_same_v<D, uint8_t>) {
            return x != S(0);
        } else {
            return (D) x;
        }
    }
};

template <typename S, typename D> int cast_launch(void *out, const ek_operand *a, size_t n) {
    Arg<S> aa;
    if (int rc = make_arg<S>(a, n, aa, "ek_hip_cast")) return rc;
    return launch_map1<CastOp<S, D>>("cast", (D *) out, n, aa);
}

template <typename S> int cast_dispatch(int dst, void *out, const ek_operand *a, size_t n) {
    switch (dst) {
        case EK_BOOL: return cast_launch<S, uint8_t>(out, a, n);
        case EK_I32: return cast_launch<S, int32_t>(out, a, n);
        case EK_U32: return cast_launch<S, uint32_t>(out, a, n);
        case EK_I64: return cast_launch<S, int64_t>(out, a, n);
        case EK_U64: return cast_launch<S, uint64_t>(out, a, n);
        case EK_F32: return cast_launch<S, float>(out, a, n);
        case EK_F64: return cast_launch<S, double>(out, a, n);
        default: return fail(EK_ERR_INVALID, "ek_hip_cast(): unknown destination type %d", dst);
    }
}

} // namespace ek

using namespace ek;

#define EK_TYPE_SWITCH(type, CALL, WHAT)                                                          \
    switch (type) {                                                                              \
        case EK_BOOL: { using T = uint8_t; return CALL; }                                         \
        case EK_I32: { using T = int32_t; return CALL; }                                          \
        case EK_U32: { using T = uint32_t; return CALL; }                                         \
        case EK_I64: { using T = int64_t; return CALL; }                                          \
        case EK_U64: { using T = uint64_t; return CALL; }                                         \
        case EK_F32: { using T = float; return CALL; }                                            \
        case EK_F64: { using T = double; return CALL; }                                           \
        default: return fail(EK_ERR_INVALID, WHAT ": unknown type %d", type);                     \
    }

#define EK_PROLOGUE(WHAT)                                                                         \
    if (int rc_ = ensure_init()) return rc_;                                                      \
    if (n == 0) return EK_OK;                                                                     \
    if (!out) return fail(EK_ERR_INVALID, WHAT ": null output pointer");

extern "C" {

int ek_hip_unary(int op, int type, void *out, const ek_operand *a, size_t n) {
    EK_PROLOGUE("ek_hip_unary()")
    EK_TYPE_SWITCH(type, unary_dispatch<T>(op, out, a, n), "ek_hip_unary()")
}

int ek_hip_binary(int op, int type, void *out, const ek_operand *a, const ek_operand *b, size_t n) {
    EK_PROLOGUE("ek_hip_binary()")
    EK_TYPE_SWITCH(type, binary_dispatch<T>(op, out, a, b, n), "ek_hip_binary()")
}

int ek_hip_ternary(int op, int type, void *out, const ek_operand *a, const ek_operand *b, const ek_operand *c,
                   size_t n) {
    EK_PROLOGUE("ek_hip_ternary()")
    EK_TYPE_SWITCH(type, ternary_dispatch<T>(op, out, a, b, c, n), "ek_hip_ternary()")
}

int ek_hip_sincos(int type, void *out, void *out_cos, const ek_operand *a, size_t n) {
    EK_PROLOGUE("ek_hip_sincos()")
    if (!out_cos) return fail(EK_ERR_INVALID, "ek_hip_sincos(): null output pointer");
    if (type == EK_F64) {
        Arg<double> ad;
        if (int rc = make_arg<double>(a, n, ad, "ek_hip_sincos")) return rc;
        return launch_map1x2<SinCosOp>("sincos", (double *) out, (double *) out_cos, n, ad);
    }
    if (type != EK_F32) return fail(EK_ERR_UNSUPPORTED, "ek_hip_sincos(): floating point types only");
    Arg<float> aa;
    if (int rc = make_arg<float>(a, n, aa, "ek_hip_sincos")) return rc;
    return launch_map1x2<SinCosOp>("sincos", (float *) out, (float *) out_cos, n, aa);
}

int ek_hip_sincosh(int type, void *out, void *out_cosh, const ek_operand *a, size_t n) {
    EK_PROLOGUE("ek_hip_sincosh()")
    if (!out_cosh) return fail(EK_ERR_INVALID, "ek_hip_sincosh(): null output pointer");
    if (type == EK_F64) {
        Arg<double> ad;
        if (int rc = make_arg<double>(a, n, ad, "ek_hip_sincosh")) return rc;
        return launch_map1x2<SinCoshOp>("sincosh", (double *) out, (double *) out_cosh, n, ad);
    }
    if (type != EK_F32) return fail(EK_ERR_UNSUPPORTED, "ek_hip_sincosh(): floating point types only");
    Arg<float> aa;
    if (int rc = make_arg<float>(a, n, aa, "ek_hip_sincosh")) return rc;
    return launch_map1x2<SinCoshOp>("sincosh", (float *) out, (float *) out_cosh, n, aa);
}

int ek_hip_compare(int op, int type, uint8_t *out, const ek_operand *a, const ek_operand *b, size_t n) {
    EK_PROLOGUE("ek_hip_compare()")
    EK_TYPE_SWITCH(type, compare_dispatch<T>(op, out, a, b, n), "ek_hip_compare()")
}

int ek_hip_select(int type, void *out, const ek_operand *mask, const ek_operand *t, const ek_operand *f, size_t n) {
    EK_PROLOGUE("ek_hip_select()")
    EK_TYPE_SWITCH(type, select_launch<T>(out, mask, t, f, n), "ek_hip_select()")
}

int ek_hip_cast(int src_type, int dst_type, void *out, const ek_operand *a, size_t n) {
    EK_PROLOGUE("ek_hip_cast()")
    int type = src_type;
    EK_TYPE_SWITCH(type, cast_dispatch<T>(dst_type, out, a, n), "ek_hip_cast()")
}

} // extern "C"
