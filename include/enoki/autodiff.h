/*
    enoki/autodiff.h -- DiffArray<Type> and Tape<Type> for eager device arrays

    Same user-facing contract as the reference's include/enoki/autodiff.h (DiffArray 127-1412, Tape
    23-124): a DiffArray is {primal value, tape index}; every differentiable operation computes the
    primal with the wrapped array type and records a node whose incoming edges carry *arrays* as
    weights; Tape::backward()/forward() sweep the recorded graph and accumulate gradients with the
    fused primitives safe_mul / safe_fmadd / hsum(safe_mul) (src/autodiff/autodiff.cpp:838-988).

    What differs, because the wrapped type is an EAGER array (every op is a kernel launch, there
    is no JIT that would drop unused values):
      * edge weights are only computed for operands that actually carry a tape index -- the
        reference computes them unconditionally and lets the JIT discard them;
      * sin/cos use the two-output sincos kernel only when the derivative is needed;
      * broadcasts of size-1 gradients are not materialised for interior nodes (kernels
        broadcast size-1 operands from a register);
      * scatter/gather offsets are kept as 32-bit arrays on the device (the reference widens
        them to Int64, autodiff.h:35).

    Tape<Type> is declared here and defined in enoki_amd/src/autodiff_impl.h; the shared library
    libenoki-hip-autodiff.so carries the explicit instantiation for HIPArray<float> (the analogue
    of src/autodiff/autodiff.cpp:1223-1241).
*/
#pragma once
#include <functional>

#include <enoki/array.h>

#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

namespace enoki {

template <typename Type> struct DiffArray;

// ---------------------------------------------------------------------------------------------
//  Fused sweep primitives with a generic fallback (src/autodiff/autodiff.cpp:1191-1221)
// ---------------------------------------------------------------------------------------------
namespace detail {
    template <typename T, typename = void> struct has_fused_safe_ops : std::false_type { };
    template <typename T>
    struct has_fused_safe_ops<T, std::void_t<decltype(T::safe_mul_(std::declval<const T &>(), std::declval<const T &>()))>>
        : std::true_type { };
}

namespace detail {
    template <typename T, typename = void> struct has_literal_test : std::false_type { };
    template <typename T>
    struct has_literal_test<T, std::void_t<decltype(std::declval<const T &>().is_literal_(scalar_t<T>(1)))>> : std::true_type { };

    /// Is `w` the host-known scalar 1?  Then safe_mul(w, g) is g itself up to the sign of zero entries
    /// (safe_mul returns +0 where g == -0), and the sweep can share g's buffer instead of streaming a copy.
    /// Backends without immediates answer false.  ENOKI_HIP_STRICT_SAFE_MUL=1 restores the literal evaluation.
    template <typename T> inline bool is_unit_weight(const T &w) {
        if constexpr (has_literal_test<T>::value) {
            static const bool strict = [] { const char *e = getenv("ENOKI_HIP_STRICT_SAFE_MUL"); return e && *e == '1'; }();
            return !strict && w.is_literal_(scalar_t<T>(1));
        } else {
            return false;
        }
    }
}

template <typename Value> inline Value safe_mul(const Value &w, const Value &g) {
    if constexpr (detail::has_fused_safe_ops<Value>::value) {
        return Value::safe_mul_(w, g);
    } else {
        Value tentative = w * g, zero_v = scalar_t<Value>(0);
        mask_t<Value> is_zero = eq(w, zero_v) | eq(g, zero_v);
        return select(is_zero, zero_v, tentative);
    }
}

template <typename Value> inline Value safe_fmadd(const Value &w, const Value &g, const Value &acc) {
    if constexpr (detail::has_fused_safe_ops<Value>::value) {
        return Value::safe_fmadd_(w, g, acc);
    } else {
        Value tentative = fmadd(w, g, acc), zero_v = scalar_t<Value>(0);
        mask_t<Value> is_zero = eq(w, zero_v) | eq(g, zero_v);
        return select(is_zero, acc, tentative);
    }
}

template <typename Value> inline Value hsum_safe_mul(const Value &w, const Value &g) {
    if constexpr (detail::has_fused_safe_ops<Value>::value)
        return Value::hsum_safe_mul_(w, g);
    else
        return hsum(safe_mul(w, g));
}

// ---------------------------------------------------------------------------------------------
//  Tape
// ---------------------------------------------------------------------------------------------
template <typename Type> struct Tape {
    using Index = uint32_t;
    using Mask = mask_t<Type>;
    using Offset = uint32_array_t<Type>;

    struct Detail;
    struct Node;
    struct Edge;
    struct Special;

    /// Global tape of this value type (single-threaded, like the reference: autodiff.cpp:207-212)
    static Tape *get();
    ~Tape();

    // ---- recording ----
    Index append(const char *label, size_t size, Index i1, const Type &w1);
    Index append(const char *label, size_t size, Index i1, Index i2, const Type &w1, const Type &w2);
    Index append(const char *label, size_t size, Index i1, Index i2, Index i3, const Type &w1, const Type &w2,
                 const Type &w3);
    Index append_node(size_t size, const char *label);
    Index append_leaf(size_t size);
    void append_edge(Index source, Index target, const Type &weight);
    void append_edge_prod(Index source, Index target, const Type &weight1, const Type &weight2);
    Index append_gather(const Offset &offset, const Mask &mask);
    void append_scatter(Index source, const Offset &offset, const Mask &mask, bool scatter_add);
    Index append_psum(Index source);
    /// A node of `size` entries whose adjoint w.r.t. `source` is supplied by the caller: backward(grad of the node) returns the
    /// contribution to grad(source).  (How a fused kernel that computed the node's value AND what its adjoint needs in one pass --
    /// enoki::vectorize-style -- enters the tape: examples/path_trace.cpp.)
    Index append_custom(Index source, size_t size, const char *label, std::function<Type(const Type &)> backward);
    Index append_reverse(Index source);
    void set_scatter_gather_operand(Index *index, size_t size, bool permute);

    // ---- reference counting (ext = held by DiffArray handles, int = held by graph edges) ----
    void inc_ref_ext(Index index);
    void dec_ref_ext(Index index);
    void inc_ref_int(Index index, Index from);
    void dec_ref_int(Index index, Index from);
    void free_node(Index index);

    // ---- sweeps ----
    void set_gradient(Index index, const Type &value, bool backward = true);
    const Type &gradient(Index index);
    void backward(Index index, bool free_graph);
    void forward(Index index, bool free_graph);
    void backward(bool free_graph);
    void forward(bool free_graph);

    // ---- housekeeping ----
    void simplify_graph();
    void set_graph_simplification(bool value);
    void set_label(Index index, const char *label);
    void push_prefix(const char *label);
    void pop_prefix();
    void set_log_level(uint32_t level);
    uint32_t log_level() const;
    std::string graphviz(const std::vector<Index> &indices);
    std::string whos() const;
    size_t node_count() const;

private:
    Tape();
    static std::unique_ptr<Tape> s_tape;
    Detail *d;
};

// ---------------------------------------------------------------------------------------------
//  DiffArray
// ---------------------------------------------------------------------------------------------
template <typename Type_> struct DiffArray : ArrayTag {
    static_assert(is_array_v<Type_> && is_dynamic_v<Type_> && array_depth_v<Type_> == 1,
                  "DiffArray requires a (non-nested) dynamic array as template parameter");

    using Type = Type_;
    using UnderlyingType = Type_;
    using Value = typename Type::Value;
    using Scalar = typename Type::Scalar;
    using ArrayType = DiffArray;
    using MaskType = DiffArray<mask_t<Type>>;
    using TapeType = enoki::Tape<Type>;
    using Index = uint32_t;
    template <typename T> using ReplaceScalar = DiffArray<replace_scalar_t<Type, T>>;
    template <typename T> using ReplaceValue = DiffArray<replace_scalar_t<Type, T>>;
    template <typename T> using ReplaceMaskValue = DiffArray<replace_scalar_t<Type, T>>;

    static constexpr size_t Depth = 1;
    static constexpr size_t Rank = Type::Rank + 1;
    static constexpr bool IsMask = is_mask_v<Type>;
    static constexpr bool IsDiff = true;
    static constexpr bool IsDynamic = true;
    static constexpr bool IsDevice = is_device_array_v<Type>;
    /// Only floating point arrays take part in differentiation (autodiff.h:143-144)
    static constexpr bool Enabled = std::is_floating_point_v<Scalar> && !IsMask;

    // -----------------------------------------------------------------------------------------
    //  Construction
    // -----------------------------------------------------------------------------------------
    DiffArray() = default;

    ~DiffArray() {
        if constexpr (Enabled) {
            if (m_index) tape()->dec_ref_ext(m_index);
        }
    }

    DiffArray(const DiffArray &a) : m_value(a.m_value), m_index(a.m_index) {
        if constexpr (Enabled) {
            if (m_index) tape()->inc_ref_ext(m_index);
        }
    }

    DiffArray(DiffArray &&a) noexcept : m_value(std::move(a.m_value)), m_index(a.m_index) { a.m_index = 0; }

    /// An expiring variable (the temporary argument of a routed unary function, array.h) lets go of its VALUE early; the tape
    /// index stays until the destructor runs
    void release_expiring_() {
        if constexpr (detail::has_release_expiring<Type>::value) m_value.release_expiring_();
    }

    DiffArray(const Type &value) : m_value(value) { }
    DiffArray(Type &&value) : m_value(std::move(value)) { }

    /// Broadcast a scalar / element list (forwarded to the value type)
    template <typename T, enable_if_t<std::is_arithmetic_v<T>> = 0>
    DiffArray(T value) : m_value(value) { }

    template <typename... Args, enable_if_t<(sizeof...(Args) > 1) && (std::is_arithmetic_v<Args> && ...)> = 0>
    DiffArray(Args... args) : m_value(args...) { }

    /// Arrays of instance pointers (vectorised method calls, array_call.h): broadcast one instance; `->` dispatches
    template <typename T = Value, enable_if_t<std::is_pointer_v<T>> = 0> DiffArray(T value) : m_value(value) { }
    template <typename T = Value, enable_if_t<std::is_pointer_v<T>> = 0> auto operator->() const { return m_value.operator->(); }

    /// Conversion between element types drops the derivative (autodiff.h:180-184)
    template <typename Type2, enable_if_t<!std::is_same_v<Type, Type2>> = 0>
    DiffArray(const DiffArray<Type2> &a) : m_value(a.value_()) { }

    template <typename Type2>
    DiffArray(const DiffArray<Type2> &a, detail::reinterpret_flag) : m_value(a.value_(), detail::reinterpret_flag()) { }

    DiffArray &operator=(const DiffArray &a) {
        if constexpr (Enabled) {
            if (a.m_index) tape()->inc_ref_ext(a.m_index);
            if (m_index) tape()->dec_ref_ext(m_index);
        }
        m_value = a.m_value;
        m_index = a.m_index;
        return *this;
    }

    DiffArray &operator=(DiffArray &&a) noexcept {
        m_value = std::move(a.m_value);
        std::swap(m_index, a.m_index);
        return *this;
    }

    // -----------------------------------------------------------------------------------------
    //  Differentiable vertical operations.  Edge weights follow autodiff.h:219-757.
    // -----------------------------------------------------------------------------------------
    DiffArray add_(const DiffArray &a) const {
        Type result = m_value + a.m_value;
        Index idx = 0;
        if constexpr (Enabled) {
            if (m_index | a.m_index)
                idx = tape()->append("add", slices(result), m_index, a.m_index, Type(1), Type(1));
        }
        return create(idx, std::move(result));
    }

    DiffArray sub_(const DiffArray &a) const {
        Type result = m_value - a.m_value;
        Index idx = 0;
        if constexpr (Enabled) {
            if (m_index | a.m_index)
                idx = tape()->append("sub", slices(result), m_index, a.m_index, Type(1), Type(-1));
        }
        return create(idx, std::move(result));
    }

    DiffArray mul_(const DiffArray &a) const {
        Type result = m_value * a.m_value;
        Index idx = 0;
        if constexpr (Enabled) {
            if (m_index | a.m_index)
                idx = tape()->append("mul", slices(result), m_index, a.m_index, a.m_value, m_value);
        }
        return create(idx, std::move(result));
    }

    DiffArray div_(const DiffArray &a) const {
        Type result = m_value / a.m_value;
        Index idx = 0;
        if constexpr (Enabled) {
            if (m_index | a.m_index) {
                Type rcp_a = rcp(a.m_value);
                Type w2 = a.m_index ? Type(-m_value * sqr(rcp_a)) : Type();
                idx = tape()->append("div", slices(result), m_index, a.m_index, rcp_a, w2);
            }
        }
        return create(idx, std::move(result));
    }

    DiffArray fmadd_(const DiffArray &a, const DiffArray &b) const {
        Type result = fmadd(m_value, a.m_value, b.m_value);
        Index idx = 0;
        if constexpr (Enabled) {
            if (m_index | a.m_index | b.m_index)
                idx = tape()->append("fmadd", slices(result), m_index, a.m_index, b.m_index, a.m_value, m_value, Type(1));
        }
        return create(idx, std::move(result));
    }

    DiffArray fmsub_(const DiffArray &a, const DiffArray &b) const {
        Type result = fmsub(m_value, a.m_value, b.m_value);
        Index idx = 0;
        if constexpr (Enabled) {
            if (m_index | a.m_index | b.m_index)
                idx = tape()->append("fmsub", slices(result), m_index, a.m_index, b.m_index, a.m_value, m_value, Type(-1));
        }
        return create(idx, std::move(result));
    }

    DiffArray fnmadd_(const DiffArray &a, const DiffArray &b) const {
        Type result = fnmadd(m_value, a.m_value, b.m_value);
        Index idx = 0;
        if constexpr (Enabled) {
            if (m_index | a.m_index | b.m_index)
                idx = tape()->append("fnmadd", slices(result), m_index, a.m_index, b.m_index,
                                     m_index ? Type(-a.m_value) : Type(), a.m_index ? Type(-m_value) : Type(), Type(1));
        }
        return create(idx, std::move(result));
    }

    DiffArray fnmsub_(const DiffArray &a, const DiffArray &b) const {
        Type result = fnmsub(m_value, a.m_value, b.m_value);
        Index idx = 0;
        if constexpr (Enabled) {
            if (m_index | a.m_index | b.m_index)
                idx = tape()->append("fnmsub", slices(result), m_index, a.m_index, b.m_index,
                                     m_index ? Type(-a.m_value) : Type(), a.m_index ? Type(-m_value) : Type(), Type(-1));
        }
        return create(idx, std::move(result));
    }

    DiffArray neg_() const {
        Index idx = 0;
        if constexpr (Enabled) {
            if (m_index) idx = tape()->append("neg", slices(m_value), m_index, Type(-1));
        }
        return create(idx, -m_value);
    }

    DiffArray abs_() const {
        Index idx = 0;
        if constexpr (Enabled) {
            if (m_index) idx = tape()->append("abs", slices(m_value), m_index, sign(m_value));
        }
        return create(idx, abs(m_value));
    }

    DiffArray sqrt_() const {
        Type result = sqrt(m_value);
        Index idx = 0;
        if constexpr (Enabled) {
            if (m_index) idx = tape()->append("sqrt", slices(result), m_index, Scalar(.5) / result);
        }
        return create(idx, std::move(result));
    }

    DiffArray rcp_() const {
        Type result = rcp(m_value);
        Index idx = 0;
        if constexpr (Enabled) {
            if (m_index) idx = tape()->append("rcp", slices(result), m_index, -sqr(result));
        }
        return create(idx, std::move(result));
    }

    DiffArray rsqrt_() const {
        Type result = rsqrt(m_value);
        Index idx = 0;
        if constexpr (Enabled) {
            if (m_index) {
                Type rsqrt_2 = sqr(result), rsqrt_3 = result * rsqrt_2;
                idx = tape()->append("rsqrt", slices(result), m_index, Scalar(-.5) * rsqrt_3);
            }
        }
        return create(idx, std::move(result));
    }

    DiffArray min_(const DiffArray &a) const {
        Type result = min(m_value, a.m_value);
        Index idx = 0;
        if constexpr (Enabled) {
            if (m_index | a.m_index) {
                mask_t<Type> m = m_value < a.m_value;
                idx = tape()->append("min", slices(result), m_index, a.m_index, select(m, Type(1), Type(0)),
                                     select(m, Type(0), Type(1)));
            }
        }
        return create(idx, std::move(result));
    }

    DiffArray max_(const DiffArray &a) const {
        Type result = max(m_value, a.m_value);
        Index idx = 0;
        if constexpr (Enabled) {
            if (m_index | a.m_index) {
                mask_t<Type> m = m_value > a.m_value;
                idx = tape()->append("max", slices(result), m_index, a.m_index, select(m, Type(1), Type(0)),
                                     select(m, Type(0), Type(1)));
            }
        }
        return create(idx, std::move(result));
    }

    static DiffArray select_(const MaskType &m, const DiffArray &t, const DiffArray &f) {
        Type result = select(m.value_(), t.m_value, f.m_value);
        Index idx = 0;
        if constexpr (Enabled) {
            if (t.m_index | f.m_index)
                idx = tape()->append("select", slices(result), t.m_index, f.m_index,
                                     select(m.value_(), Type(1), Type(0)), select(m.value_(), Type(0), Type(1)));
        }
        return create(idx, std::move(result));
    }

    DiffArray sin_() const {
        if constexpr (Enabled) {
            if (m_index) {
                auto [s, c] = sincos(m_value);
                Index idx = tape()->append("sin", slices(m_value), m_index, c);
                return create(idx, std::move(s));
            }
        }
        return create(0, sin(m_value));
    }

    DiffArray cos_() const {
        if constexpr (Enabled) {
            if (m_index) {
                auto [s, c] = sincos(m_value);
                Index idx = tape()->append("cos", slices(m_value), m_index, -s);
                return create(idx, std::move(c));
            }
        }
        return create(0, cos(m_value));
    }

    std::pair<DiffArray, DiffArray> sincos_() const {
        auto [s, c] = sincos(m_value);
        Index idx_s = 0, idx_c = 0;
        if constexpr (Enabled) {
            if (m_index) {
                idx_s = tape()->append("sin", slices(m_value), m_index, c);
                idx_c = tape()->append("cos", slices(m_value), m_index, -s);
            }
        }
        return { create(idx_s, std::move(s)), create(idx_c, std::move(c)) };
    }

    DiffArray exp_() const {
        Type result = exp(m_value);
        Index idx = 0;
        if constexpr (Enabled) {
            if (m_index) idx = tape()->append("exp", slices(m_value), m_index, result);
        }
        return create(idx, std::move(result));
    }

    DiffArray log_() const {
        Index idx = 0;
        if constexpr (Enabled) {
            if (m_index) idx = tape()->append("log", slices(m_value), m_index, rcp(m_value));
        }
        return create(idx, log(m_value));
    }

    // ---- second wave (reference autodiff.h:366-377, 532-732): value from the fused kernel, edge weight
    //      from the same derivative expression the reference records -------------------------------------
#define ENOKI_HIP_DIFF_UNARY(name, label, value_expr, weight_expr)                                   \
    DiffArray name##_() const {                                                                      \
        Type result = value_expr;                                                                    \
        Index idx = 0;                                                                               \
        if constexpr (Enabled) {                                                                     \
            if (m_index) idx = tape()->append(label, slices(m_value), m_index, weight_expr);         \
        }                                                                                            \
        return create(idx, std::move(result));                                                       \
    }

    ENOKI_HIP_DIFF_UNARY(tan,   "tan",   tan(m_value),   sec_sqr(m_value))
    ENOKI_HIP_DIFF_UNARY(cot,   "cot",   cot(m_value),   -sqr(csc(m_value)))
    ENOKI_HIP_DIFF_UNARY(csc,   "csc",   csc(m_value),   -result * cot(m_value))
    ENOKI_HIP_DIFF_UNARY(sec,   "sec",   sec(m_value),   result * tan(m_value))
    ENOKI_HIP_DIFF_UNARY(asin,  "asin",  asin(m_value),  rsqrt(Type(1) - sqr(m_value)))
    ENOKI_HIP_DIFF_UNARY(acos,  "acos",  acos(m_value),  -rsqrt(Type(1) - sqr(m_value)))
    ENOKI_HIP_DIFF_UNARY(atan,  "atan",  atan(m_value),  rcp_1p_sqr(m_value))
    ENOKI_HIP_DIFF_UNARY(csch,  "csch",  csch(m_value),  -result * coth(m_value))
    ENOKI_HIP_DIFF_UNARY(sech,  "sech",  sech(m_value),  -result * tanh(m_value))
    ENOKI_HIP_DIFF_UNARY(tanh,  "tanh",  tanh(m_value),  sech_sqr(m_value))
    ENOKI_HIP_DIFF_UNARY(asinh, "asinh", asinh(m_value), rsqrt(Type(1) + sqr(m_value)))
    ENOKI_HIP_DIFF_UNARY(acosh, "acosh", acosh(m_value), rsqrt(sqr(m_value) - Type(1)))
    ENOKI_HIP_DIFF_UNARY(atanh, "atanh", atanh(m_value), rcp(Type(1) - sqr(m_value)))
    ENOKI_HIP_DIFF_UNARY(cbrt,  "cbrt",  cbrt(m_value),  Type(1.f) / (Type(3) * sqr(result)))
#undef ENOKI_HIP_DIFF_UNARY

    // sinh / cosh (autodiff.h:635-657: value and weight are the two halves of ONE sincosh).  An array type that leaves unary maps
    // unevaluated (HIPArray) takes the two halves as two maps of the same argument -- the same bits (sinh(x) and cosh(x) are what
    // sincosh(x) returns, array_math.h:997-1127), but a reduction of the value then reads THROUGH the argument and the weight is
    // formed by whoever consumes it (a chain, the sweep's product) instead of being written next to the value.
    DiffArray sinh_() const {
        if constexpr (detail::has_sec_sqr<Type>::value) {
            Type s = sinh(m_value);
            Index idx = 0;
            if constexpr (Enabled) {
                if (m_index) idx = tape()->append("sinh", slices(m_value), m_index, cosh(m_value));
            }
            return create(idx, std::move(s));
        } else {
            auto [s, c] = sincosh(m_value);
            Index idx = 0;
            if constexpr (Enabled) {
                if (m_index) idx = tape()->append("sinh", slices(m_value), m_index, c);
            }
            return create(idx, std::move(s));
        }
    }

    DiffArray cosh_() const {
        if constexpr (detail::has_sec_sqr<Type>::value) {
            Type c = cosh(m_value);
            Index idx = 0;
            if constexpr (Enabled) {
                if (m_index) idx = tape()->append("cosh", slices(m_value), m_index, sinh(m_value));
            }
            return create(idx, std::move(c));
        } else {
            auto [s, c] = sincosh(m_value);
            Index idx = 0;
            if constexpr (Enabled) {
                if (m_index) idx = tape()->append("cosh", slices(m_value), m_index, s);
            }
            return create(idx, std::move(c));
        }
    }

    std::pair<DiffArray, DiffArray> sincosh_() const {
        auto [s, c] = sincosh(m_value);
        Index idx_s = 0, idx_c = 0;
        if constexpr (Enabled) {
            if (m_index) {
                idx_s = tape()->append("sinh", slices(m_value), m_index, c);
                idx_c = tape()->append("cosh", slices(m_value), m_index, s);
            }
        }
        return { create(idx_s, std::move(s)), create(idx_c, std::move(c)) };
    }

    /// atan2(y = *this, x) (autodiff.h:618-633)
    DiffArray atan2_(const DiffArray &x) const {
        Index idx = 0;
        if constexpr (Enabled) {
            if (m_index | x.m_index) {
                Type il2 = rcp(sqr(m_value) + sqr(x.m_value));
                idx = tape()->append("atan2", slices(il2), m_index, x.m_index, il2 * x.m_value, -il2 * m_value);
            }
        }
        return create(idx, atan2(m_value, x.m_value));
    }

    /// value & mask: gradient flows where the mask is set (autodiff.h:780-787)
    template <typename T = Type, enable_if_t<!is_mask_v<T>> = 0>
    DiffArray and_(const MaskType &m) const {
        Index idx = 0;
        if constexpr (Enabled) {
            if (m_index) idx = tape()->append("and", slices(m_value), m_index, select(m.value_(), Type(1), Type(0)));
        }
        return create(idx, m_value & m.value_());
    }

    template <typename T = Type, enable_if_t<!is_mask_v<T>> = 0>
    DiffArray or_(const MaskType &m) const {
        Index idx = 0;
        if constexpr (Enabled) {
            if (m_index) idx = tape()->append("or", slices(m_value), m_index, Type(1));
        }
        return create(idx, m_value | m.value_());
    }

    // -----------------------------------------------------------------------------------------
    //  Operations without derivatives (autodiff.h:455-489, 759-953)
    // -----------------------------------------------------------------------------------------
    DiffArray and_(const DiffArray &a) const { return create(0, m_value & a.m_value); }
    DiffArray or_(const DiffArray &a) const { return create(0, m_value | a.m_value); }
    DiffArray xor_(const DiffArray &a) const { return create(0, m_value ^ a.m_value); }
    DiffArray not_() const { return create(0, ~m_value); }
    DiffArray mod_(const DiffArray &a) const { return create(0, m_value % a.m_value); }
    DiffArray mulhi_(const DiffArray &a) const { return create(0, mulhi(m_value, a.m_value)); }
    DiffArray sl_(const DiffArray &a) const { return create(0, m_value << a.m_value); }
    DiffArray sr_(const DiffArray &a) const { return create(0, m_value >> a.m_value); }
    DiffArray floor_() const { return create(0, floor(m_value)); }
    DiffArray ceil_() const { return create(0, ceil(m_value)); }
    DiffArray round_() const { return create(0, round(m_value)); }
    DiffArray trunc_() const { return create(0, trunc(m_value)); }
    DiffArray popcnt_() const { return create(0, popcnt(m_value)); }
    DiffArray lzcnt_() const { return create(0, lzcnt(m_value)); }
    DiffArray tzcnt_() const { return create(0, tzcnt(m_value)); }
    DiffArray sign_() const { return create(0, sign(m_value)); }
    // the remaining gradient-free members of the reference's DiffArray (autodiff.h:480-489, 803-815, 847-852, 920-947)
    template <typename T> T floor2int_() const { return T(floor2int<typename T::UnderlyingType>(m_value)); }
    template <typename T> T ceil2int_() const { return T(ceil2int<typename T::UnderlyingType>(m_value)); }
    DiffArray andnot_(const DiffArray &a) const { return create(0, m_value & ~a.m_value); }
    DiffArray rol_(const DiffArray &a) const { return create(0, rol(m_value, a.m_value)); }
    DiffArray ror_(const DiffArray &a) const { return create(0, ror(m_value, a.m_value)); }
    /// the first active entry (no gradient is carried by the returned scalar)
    Scalar extract_(const MaskType &mask) const { return extract(m_value, mask.value_()); }

    MaskType eq_(const DiffArray &d) const { return MaskType(eq(m_value, d.m_value)); }
    MaskType neq_(const DiffArray &d) const { return MaskType(neq(m_value, d.m_value)); }
    MaskType lt_(const DiffArray &d) const { return MaskType(m_value < d.m_value); }
    MaskType le_(const DiffArray &d) const { return MaskType(m_value <= d.m_value); }
    MaskType gt_(const DiffArray &d) const { return MaskType(m_value > d.m_value); }
    MaskType ge_(const DiffArray &d) const { return MaskType(m_value >= d.m_value); }

    // -----------------------------------------------------------------------------------------
    //  Scatter / gather with array operands (autodiff.h:962-998, array_struct.h:9-123)
    // -----------------------------------------------------------------------------------------
    template <bool IsPermute, typename Index_>
    static DiffArray gather_array_(const DiffArray &source, const Index_ &index, const MaskType &mask) {
        if (source.size() <= 1)
            return source & mask;
        Type result = Type::template gather_array_<IsPermute>(source.m_value, detach(index), mask.value_());
        Index idx = 0;
        if constexpr (Enabled) {
            if (source.m_index) {
                auto *t = tape();
                t->set_scatter_gather_operand(const_cast<Index *>(&source.m_index), source.size(), IsPermute);
                idx = t->append_gather(typename TapeType::Offset(detach(index)), mask.value_());
                t->set_scatter_gather_operand(nullptr, 0, false);
            }
        }
        return create(idx, std::move(result));
    }

    /// Structure-of-arrays gather (Array<DiffArray, N> sources) when the backend looks the N components up as ONE record
    /// per element (HIPArray::gather_records_): the values come from that kernel, the tape gets the same N gather nodes as
    /// from N calls of gather_array_ -- their adjoints share the index array and run as one multi-table scatter_add.
    template <size_t N, typename Index_>
    static bool gather_multi_(const DiffArray *sources, DiffArray *results, const Index_ &index, const MaskType &mask) {
        if constexpr (!detail::has_gather_records<Type, std::decay_t<decltype(detach(index))>>::value) {
            return false;
        } else {
            Type src[N], res[N];
            for (size_t c = 0; c < N; ++c) src[c] = sources[c].m_value;
            if (!Type::template gather_records_<N>(src, res, detach(index), mask.value_()))
                return false;
            for (size_t c = 0; c < N; ++c) {
                Index idx = 0;
                if constexpr (Enabled) {
                    if (sources[c].m_index) {
                        auto *t = tape();
                        t->set_scatter_gather_operand(const_cast<Index *>(&sources[c].m_index), sources[c].size(), false);
                        idx = t->append_gather(typename TapeType::Offset(detach(index)), mask.value_());
                        t->set_scatter_gather_operand(nullptr, 0, false);
                    }
                }
                results[c] = create(idx, std::move(res[c]));
            }
            return true;
        }
    }

    template <bool IsPermute, typename Index_>
    static void scatter_array_(DiffArray &target, const DiffArray &value, const Index_ &index, const MaskType &mask) {
        Type::template scatter_array_<IsPermute>(target.m_value, value.m_value, detach(index), mask.value_());
        if constexpr (Enabled) {
            if (target.m_index | value.m_index) {
                auto *t = tape();
                t->set_scatter_gather_operand(&target.m_index, target.size(), IsPermute);
                t->append_scatter(value.m_index, typename TapeType::Offset(detach(index)), mask.value_(), false);
                t->set_scatter_gather_operand(nullptr, 0, false);
            }
        }
    }

    template <bool IsPermute, typename Index_>
    static void scatter_add_array_(DiffArray &target, const DiffArray &value, const Index_ &index, const MaskType &mask) {
        Type::template scatter_add_array_<IsPermute>(target.m_value, value.m_value, detach(index), mask.value_());
        if constexpr (Enabled) {
            if (target.m_index | value.m_index) {
                auto *t = tape();
                t->set_scatter_gather_operand(&target.m_index, target.size(), IsPermute);
                t->append_scatter(value.m_index, typename TapeType::Offset(detach(index)), mask.value_(), true);
                t->set_scatter_gather_operand(nullptr, 0, false);
            }
        }
    }

    // -----------------------------------------------------------------------------------------
    //  Horizontal operations (autodiff.h:1007-1096)
    // -----------------------------------------------------------------------------------------
    bool all_() const { return all(m_value); }
    bool any_() const { return any(m_value); }
    size_t count_() const { return count(m_value); }

    DiffArray hsum_() const {
        Index idx = 0;
        if constexpr (Enabled) {
            if (m_index) idx = tape()->append("hsum", 1, m_index, Type(1));
        }
        return create(idx, hsum(m_value));
    }

    DiffArray hprod_() const {
        Type result = hprod(m_value);
        Index idx = 0;
        if constexpr (Enabled) {
            if (m_index)
                idx = tape()->append("hprod", 1, m_index,
                                     select(eq(m_value, Type(0)), Type(0), result / m_value));
        }
        return create(idx, std::move(result));
    }

    DiffArray hmax_() const {
        if (Enabled && m_index != 0)
            throw std::runtime_error("DiffArray::hmax_(): gradients not implemented (as in the reference)");
        return create(0, hmax(m_value));
    }

    DiffArray hmin_() const {
        if (Enabled && m_index != 0)
            throw std::runtime_error("DiffArray::hmin_(): gradients not implemented (as in the reference)");
        return create(0, hmin(m_value));
    }

    DiffArray psum_() const {
        Index idx = 0;
        if constexpr (Enabled) {
            if (m_index) idx = tape()->append_psum(m_index);
        }
        return create(idx, psum(m_value));
    }

    DiffArray reverse_() const {
        Index idx = 0;
        if constexpr (Enabled) {
            if (m_index) idx = tape()->append_reverse(m_index);
        }
        return create(idx, reverse(m_value));
    }

    // -----------------------------------------------------------------------------------------
    //  Initializers (forwarded), storage access
    // -----------------------------------------------------------------------------------------
    static DiffArray empty_(size_t size) { return Type::empty_(size); }
    static DiffArray zero_(size_t size) { return Type::zero_(size); }
    static DiffArray full_(const Scalar &value, size_t size) { return Type::full_(value, size); }
    static DiffArray arange_(ptrdiff_t start, ptrdiff_t stop, ptrdiff_t step) { return Type::arange_(start, stop, step); }
    static DiffArray linspace_(Scalar min, Scalar max, size_t size) { return Type::linspace_(min, max, size); }
    static DiffArray map(void *ptr, size_t size, bool dealloc = false) { return Type::map(ptr, size, dealloc); }
    static DiffArray copy(const void *ptr, size_t size) { return Type::copy(ptr, size); }

    size_t size() const { return m_value.size(); }
    size_t slices_() const { return m_value.size(); }
    bool empty() const { return m_value.empty(); }
    void resize(size_t size) { m_value.resize(size); }
    void set_slices_(size_t size) { m_value.set_slices_(size); }
    const Scalar *data() const { return m_value.data(); }
    Scalar *data() { return m_value.data(); }
    Scalar coeff(size_t i) const { return m_value.coeff(i); }
    Scalar operator[](size_t i) const { return m_value.coeff(i); }
    auto operator[](const MaskType &mask) { return masked(*this, mask); }
    DiffArray &eval() { m_value.eval(); return *this; }
    const DiffArray &eval() const { m_value.eval(); return *this; }
    DiffArray &managed() { m_value.managed(); return *this; }

    Index index_() const { return m_index; }
    Type &value_() { return m_value; }
    const Type &value_() const { return m_value; }

    void set_index_(Index index) {
        if constexpr (Enabled) {
            if (index) tape()->inc_ref_ext(index);
            if (m_index) tape()->dec_ref_ext(m_index);
        }
        m_index = index;
    }

    // -----------------------------------------------------------------------------------------
    //  Autodiff interface (autodiff.h:1189-1347)
    // -----------------------------------------------------------------------------------------
    void set_requires_gradient_(bool value) {
        static_assert(Enabled, "set_requires_gradient(): floating point arrays only");
        if (value && m_index == 0) {
            m_index = tape()->append_leaf(slices(m_value));
        } else if (!value && m_index != 0) {
            tape()->dec_ref_ext(m_index);
            m_index = 0;
        }
    }
    bool requires_gradient_() const { return Enabled && m_index != 0; }

    const Type &gradient_() const { return tape()->gradient(m_index); }
    static const Type &gradient_static_(Index index) { return tape()->gradient(index); }
    void set_gradient_(const Type &value, bool backward = true) { tape()->set_gradient(m_index, value, backward); }
    void backward_(bool free_graph) const { tape()->backward(m_index, free_graph); }
    void forward_(bool free_graph) const { tape()->forward(m_index, free_graph); }
    static void backward_static_(bool free_graph) { tape()->backward(free_graph); }
    static void forward_static_(bool free_graph) { tape()->forward(free_graph); }
    void set_label_(const char *label) const {
        if constexpr (Enabled) tape()->set_label(m_index, label);
    }
    static std::string graphviz_(const std::vector<Index> &indices) { return tape()->graphviz(indices); }
    static void push_prefix_(const char *label) { if constexpr (Enabled) tape()->push_prefix(label); }
    static void pop_prefix_() { if constexpr (Enabled) tape()->pop_prefix(); }
    static void set_log_level_(uint32_t level) { if constexpr (Enabled) tape()->set_log_level(level); }
    static uint32_t log_level_() { if constexpr (Enabled) return tape()->log_level(); else return 0; }
    static void set_graph_simplification_(bool value) { if constexpr (Enabled) tape()->set_graph_simplification(value); }
    static void simplify_graph_() { if constexpr (Enabled) tape()->simplify_graph(); }
    static std::string whos_() { return tape()->whos(); }
    static void inc_ref_ext_(Index index) { if constexpr (Enabled) tape()->inc_ref_ext(index); }
    static void dec_ref_ext_(Index index) { if constexpr (Enabled) tape()->dec_ref_ext(index); }

    /// value = f(source) computed outside the tape, with `backward` as its adjoint (Tape::append_custom)
    static DiffArray custom_(const DiffArray &source, Type &&value, const char *label, std::function<Type(const Type &)> backward) {
        Index idx = 0;
        if constexpr (Enabled) {
            if (source.m_index) idx = tape()->append_custom(source.m_index, slices(value), label, std::move(backward));
        }
        return create(idx, std::move(value));
    }

    static DiffArray create(Index index, Type &&value) {
        DiffArray result(std::move(value));
        result.m_index = index;
        return result;
    }

private:
    static TapeType *tape() { return TapeType::get(); }

    Type m_value;
    Index m_index = 0;
};

// ---------------------------------------------------------------------------------------------
//  Free functions (autodiff.h:1414-1531)
// ---------------------------------------------------------------------------------------------
namespace detail {
    template <typename T> struct is_static_array : std::false_type { };
    template <typename V, size_t N> struct is_static_array<Array<V, N>> : std::true_type { };
    template <typename T> constexpr bool is_static_array_v = is_static_array<std::decay_t<T>>::value;
}

/// Nested types (Array<DiffArray<...>, N>) are handled component by component (autodiff.h:1414-1500)
template <typename T> inline bool requires_gradient(const T &a) {
    if constexpr (detail::is_static_array_v<T>) {
        bool result = false;
        for (size_t i = 0; i < T::Size; ++i) result = result || requires_gradient(a.coeff(i));
        return result;
    } else if constexpr (is_diff_array_v<T>) {
        return a.requires_gradient_();
    } else {
        return false;
    }
}

template <typename T> inline void set_requires_gradient(T &a, bool value = true) {
    if constexpr (detail::is_static_array_v<T>) {
        for (size_t i = 0; i < T::Size; ++i) set_requires_gradient(a.coeff(i), value);
    } else if constexpr (is_diff_array_v<T>) {
        a.set_requires_gradient_(value);
    }
}

template <typename T> inline decltype(auto) gradient(const T &a) {
    if constexpr (detail::is_static_array_v<T>) {
        using G = std::decay_t<decltype(gradient(a.coeff(0)))>;
        Array<G, T::Size> result;
        for (size_t i = 0; i < T::Size; ++i) result.coeff(i) = gradient(a.coeff(i));
        return result;
    } else {
        return a.gradient_();
    }
}
template <typename T> inline uint32_t gradient_index(const T &a) { return a.index_(); }
template <typename T1, typename T2> inline void set_gradient(T1 &a, const T2 &b, bool backward = true) {
    a.set_gradient_(typename T1::Type(b), backward);
}
template <typename T> inline void reattach(T &a, const T &b) { a.set_index_(b.index_()); }
template <typename T> inline void backward(const T &a, bool free_graph = true) { a.backward_(free_graph); }
template <typename T> inline void forward(const T &a, bool free_graph = true) { a.forward_(free_graph); }
template <typename T> inline void backward(bool free_graph = true) { T::backward_static_(free_graph); }
template <typename T> inline void forward(bool free_graph = true) { T::forward_static_(free_graph); }
template <typename T, enable_if_t<is_diff_array_v<T>> = 0> inline void set_label(const T &a, const char *label) {
    a.set_label_(label);
}
template <typename T> inline std::string graphviz(const T &a) {
    std::vector<uint32_t> indices;
    if (a.index_()) indices.push_back(a.index_());
    return T::graphviz_(indices);
}

} // namespace enoki
