/*
    integration/enoki/hip.h -- the header a maintainer of the REFERENCE adds next to its include/enoki/cuda.h

    This file is written against the reference's OWN headers (`#include <enoki/array.h>` below resolves to
    /root/reference/include/enoki/array.h: compile with `-I<reference>/include -Iintegration -Iinclude`), not against
    this repository's slim array.h.  It defines `enoki::HIPArray<Value>` as one more array backend behind the reference's
    ArrayBase / array_router.h / array_math.h / array_struct.h: the member concept of CUDAArray<Value> (cuda.h:205-954),
    every member forwarding to ONE entry point of the C ABI (include/enoki_hip.h -> libenoki-hip.so).  Nothing in the
    reference tree changes; the four JIT hooks that array_struct.h / array_router.h call on "CUDA" arrays
    (cuda_eval, cuda_sync, cuda_set_scatter_gather_operand, cuda_var_mark_dirty: cuda.h:53-170) are no-ops for an eager
    backend and are defined in integration/hip_hooks.cpp.

    Two pieces of bookkeeping let the reference's autodiff layer run at a sensible speed without an edit (see
    hip_detail::Buffer): a float array remembers that it is the product / fused multiply-add of two (three) others, so
    that the trace fragments with which autodiff.cpp spells safe_mul / safe_fmadd become one fused kernel
    (integration/hip_hooks.cpp), and a 64-bit integer array remembers the 32-bit array it was widened from, so that
    gather_ / scatter_add_ pass the original to the library.  Both tags expire when an operand hands out a mutable pointer.

    tests/cpp/reference_side_hip.cpp instantiates the same templated functions on the reference's CPU arrays
    (DynamicArray<Packet<float>>) and on this class, in ONE binary, and compares the results on the device box.
*/
#pragma once

#include <enoki/array.h>
#include <enoki/cuda.h>          // declarations of the hooks named above (is_cuda_array_v lives in array_traits.h)
#include <enoki_hip.h>

#include <cstring>
#include <map>
#include <memory>
#include <vector>
#include <stdexcept>
#include <string>
#include <utility>

NAMESPACE_BEGIN(enoki)

NAMESPACE_BEGIN(hip_detail)

inline void check(int rc, const char *what) {
    if (rc != EK_OK)
        throw std::runtime_error(std::string("HIPArray::") + what + "(): " + ek_hip_last_error());
}

template <typename T> constexpr int type_code() {
    if constexpr (std::is_same_v<T, bool>) return EK_BOOL;
    else if constexpr (std::is_same_v<T, float>) return EK_F32;
    else if constexpr (std::is_same_v<T, double>) return EK_F64;
    else if constexpr (std::is_integral_v<T> && sizeof(T) == 4) return std::is_signed_v<T> ? EK_I32 : EK_U32;
    else if constexpr (std::is_integral_v<T> && sizeof(T) == 8) return std::is_signed_v<T> ? EK_I64 : EK_U64;
    else if constexpr (std::is_pointer_v<T>) return EK_U64;        // arrays of instance pointers (array_call.h)
    else return -1;
}

/// one device allocation, shared by the handles that copy it (value semantics: copies share, writers are scatter / data())
struct Buffer {
    void *ptr = nullptr;
    size_t size = 0;
    bool owned = true;

    /// What a float array was computed from, when that was a plain product or fused multiply-add (set by mul_ / fmadd_):
    /// lets integration/hip_hooks.cpp recognise the reference's spelling of safe_mul / safe_fmadd (autodiff.cpp:1191-1221)
    /// and run ONE fused kernel for it.  Weak: a tag never keeps an operand alive.
    std::weak_ptr<Buffer> prod[3];
    uint64_t prod_version[3] = { 0, 0, 0 };     // the operands' `version` when the tag was made
    int prod_kind = 0;                          // 0: unknown, 1: prod[0] * prod[1], 2: fma(prod[0], prod[1], prod[2])

    /// A 64-bit integer array that was widened from a 32-bit one (the reference's tape turns every gather offset into Int64,
    /// autodiff.cpp:355-366): gather_ / scatter_add_ hand the ORIGINAL to the library while it is alive and unmodified --
    /// the kernels are written for 32-bit indices and would otherwise narrow the array again (12 bytes per element).
    std::weak_ptr<Buffer> narrow_src;
    int narrow_code = 0;
    uint64_t narrow_version = 0;
    uint64_t version = 0;                       // bumped whenever a mutable pointer is handed out

    /// A mask that is still a recipe: 1: (lazy[0] == 0), 2: (lazy[0] == 0) | (lazy[1] == 0) over float arrays.  The trace
    /// fragments of safe_mul produce exactly these; they are evaluated only if somebody reads the mask (get()).
    std::shared_ptr<Buffer> lazy[2];
    int lazy_kind = 0;

    /// the device pointer of an INPUT (evaluates a pending recipe first)
    void *get() {
        if (lazy_kind != 0) materialize();
        return ptr;
    }
    void materialize() {
        const int kind = lazy_kind;
        lazy_kind = 0;
        check(ek_hip_malloc(size ? size : 1, &ptr), "mask recipe");
        ek_operand zero{ nullptr, 0, 1 }, a{ lazy[0]->get(), 0, lazy[0]->size };
        if (kind == 1) {
            check(ek_hip_compare(EK_EQ, EK_F32, (uint8_t *) ptr, &a, &zero, size), "mask recipe");
        } else {
            void *tmp = nullptr;
            check(ek_hip_malloc(lazy[0]->size ? lazy[0]->size : 1, &tmp), "mask recipe");
            check(ek_hip_compare(EK_EQ, EK_F32, (uint8_t *) tmp, &a, &zero, lazy[0]->size), "mask recipe");
            ek_operand b{ lazy[1]->get(), 0, lazy[1]->size };
            void *tmp2 = nullptr;
            check(ek_hip_malloc(lazy[1]->size ? lazy[1]->size : 1, &tmp2), "mask recipe");
            check(ek_hip_compare(EK_EQ, EK_F32, (uint8_t *) tmp2, &b, &zero, lazy[1]->size), "mask recipe");
            ek_operand ea{ tmp, 0, lazy[0]->size }, eb{ tmp2, 0, lazy[1]->size };
            check(ek_hip_binary(EK_OR, EK_BOOL, ptr, &ea, &eb, size), "mask recipe");
            ek_hip_free(tmp); ek_hip_free(tmp2);              // stream-ordered reuse
        }
        lazy[0].reset(); lazy[1].reset();
    }
    ~Buffer() { if (owned && ptr) ek_hip_free(ptr); }
};

/// The reference's autodiff.cpp spells safe_mul / safe_fmadd for "CUDA" arrays as three trace fragments on variable
/// indices (autodiff.cpp:1198-1202, 1214-1218).  An eager backend has no variables; index_() therefore parks the buffer in a
/// small ring of recently named buffers and from_index_() picks it up again -- enough for integration/hip_hooks.cpp to
/// execute those fragments as kernels (a maintainer would rather replace the 30 lines by Value::safe_mul_()).
struct Handles {
    static constexpr uint32_t kSlots = 256;
    std::shared_ptr<Buffer> slot[kSlots];
    uint32_t next = 1;
    static Handles &get() { static Handles h; return h; }
    uint32_t park(const std::shared_ptr<Buffer> &b) { uint32_t id = next++; slot[id % kSlots] = b; return id; }
    std::shared_ptr<Buffer> find(uint32_t id) const {
        if (id == 0 || id + kSlots < next) throw std::runtime_error("HIPArray: stale variable index");
        return slot[id % kSlots];
    }
};

/// The array that the next gather / scatter / scatter_add addresses: array_struct.h announces it through
/// cuda_set_scatter_gather_operand(array.index_()) before it hands the bare data() pointer to the member (and withdraws it
/// afterwards).  The eager backend uses the announcement to recover the table SIZE, which the member concept does not
/// carry: ek_hip_scatter_add needs it for its LDS-binned path (8x the device atomics on large inputs).
struct Operand { const void *ptr = nullptr; size_t size = 0; };
inline Operand &announced_operand() { static thread_local Operand o; return o; }

NAMESPACE_END(hip_detail)

template <typename Value>
struct HIPArray : ArrayBase<value_t<Value>, HIPArray<Value>> {
    template <typename T> friend struct HIPArray;
    using Index = uint32_t;

    static constexpr EnokiType Type = enoki_type_v<Value>;
    static constexpr bool IsCUDA = true;        // "device array": array_math.h / array_router.h skip their CPU packet loops
    static constexpr int Code = hip_detail::type_code<Value>();
    template <typename T> using ReplaceValue = HIPArray<T>;
    using MaskType = HIPArray<bool>;
    using ArrayType = HIPArray;

    // ---- construction ------------------------------------------------------------------------------------------------
    HIPArray() = default;
    HIPArray(const HIPArray &) = default;
    HIPArray(HIPArray &&) = default;
    HIPArray &operator=(const HIPArray &) = default;
    HIPArray &operator=(HIPArray &&) = default;

    HIPArray(Value value) {                                             // scalar -> size-1 array (cuda.h:263-313)
        allocate(1);
        uint64_t bits = 0;
        memcpy(&bits, &value, sizeof(Value));
        hip_detail::check(ek_hip_fill(Code, m_buf->ptr, bits, 1), "HIPArray");
    }
    template <typename T, enable_if_t<std::is_scalar_v<T> && !std::is_same_v<T, Value>> = 0>
    HIPArray(T value) : HIPArray((Value) value) { }

    template <typename T> HIPArray(const HIPArray<T> &v) {              // conversion (cuda.h:236-247)
        if (!v.m_buf) return;
        allocate(v.size());
        ek_operand a = v.operand();
        hip_detail::check(ek_hip_cast(HIPArray<T>::Code, Code, m_buf->ptr, &a, size()), "HIPArray(cast)");
        if constexpr (std::is_integral_v<T> && sizeof(T) == 4 && std::is_integral_v<Value> && sizeof(Value) == 8 &&
                      !std::is_same_v<T, bool>) {
            m_buf->narrow_src = v.m_buf;
            m_buf->narrow_code = HIPArray<T>::Code;
            m_buf->narrow_version = v.m_buf->version;
        }
    }

    /// index operand for the library: the 32-bit array this one was widened from, if that is still valid
    ek_operand index_operand(int &code) const {
        code = Code;
        if (m_buf && m_buf->narrow_code != 0) {
            if (auto src = m_buf->narrow_src.lock(); src && src->version == m_buf->narrow_version && src->size == m_buf->size) {
                code = m_buf->narrow_code;
                return ek_operand{ src->get(), 0, src->size };
            }
        }
        return operand();
    }
    template <typename T> HIPArray(const HIPArray<T> &v, detail::reinterpret_flag) {   // same bits (cuda.h:249-258)
        static_assert(sizeof(T) == sizeof(Value));
        m_buf = v.m_buf;
    }
    template <typename T, enable_if_t<std::is_scalar_v<T>> = 0>
    HIPArray(const T &value, detail::reinterpret_flag) : HIPArray(memcpy_cast<Value>(value)) { }

    template <typename... Args, enable_if_t<(sizeof...(Args) > 1)> = 0> HIPArray(Args &&... args) {   // element list (cuda.h:319-323)
        Value data[] = { (Value) args... };
        *this = copy(data, sizeof...(Args));
    }

    // ---- vertical operations: one C-ABI call each --------------------------------------------------------------------
#define ENOKI_HIP_UNARY(name, code)                                                                                    \
    HIPArray name##_() const {                                                                                         \
        HIPArray r = empty_(size());                                                                                   \
        ek_operand a = operand();                                                                                      \
        hip_detail::check(ek_hip_unary(code, Code, r.m_buf->ptr, &a, r.size()), #name);                                \
        return r;                                                                                                      \
    }
#define ENOKI_HIP_BINARY(name, code)                                                                                   \
    HIPArray name##_(const HIPArray &v) const {                                                                        \
        HIPArray r = empty_(broadcast(size(), v.size()));                                                              \
        ek_operand a = operand(), b = v.operand();                                                                     \
        hip_detail::check(ek_hip_binary(code, Code, r.m_buf->ptr, &a, &b, r.size()), #name);                           \
        if ((code) == EK_MUL && Code == EK_F32 && m_buf && v.m_buf) {                                                  \
            r.m_buf->prod_kind = 1; r.m_buf->prod[0] = m_buf; r.m_buf->prod[1] = v.m_buf;                              \
            r.m_buf->prod_version[0] = m_buf->version; r.m_buf->prod_version[1] = v.m_buf->version;                    \
        }                                                                                                              \
        return r;                                                                                                      \
    }
#define ENOKI_HIP_TERNARY(name, code)                                                                                  \
    HIPArray name##_(const HIPArray &v, const HIPArray &w) const {                                                     \
        HIPArray r = empty_(broadcast(broadcast(size(), v.size()), w.size()));                                         \
        ek_operand a = operand(), b = v.operand(), c = w.operand();                                                    \
        hip_detail::check(ek_hip_ternary(code, Code, r.m_buf->ptr, &a, &b, &c, r.size()), #name);                      \
        if ((code) == EK_FMADD && Code == EK_F32 && m_buf && v.m_buf && w.m_buf) {                                      \
            r.m_buf->prod_kind = 2; r.m_buf->prod[0] = m_buf; r.m_buf->prod[1] = v.m_buf; r.m_buf->prod[2] = w.m_buf;  \
            r.m_buf->prod_version[0] = m_buf->version; r.m_buf->prod_version[1] = v.m_buf->version;                    \
            r.m_buf->prod_version[2] = w.m_buf->version;                                                               \
        }                                                                                                              \
        return r;                                                                                                      \
    }
#define ENOKI_HIP_COMPARE(name, code)                                                                                  \
    MaskType name##_(const HIPArray &v) const {                                                                        \
        MaskType r = MaskType::empty_(broadcast(size(), v.size()));                                                    \
        ek_operand a = operand(), b = v.operand();                                                                     \
        hip_detail::check(ek_hip_compare(code, Code, (uint8_t *) r.m_buf->ptr, &a, &b, r.size()), #name);              \
        return r;                                                                                                      \
    }
    ENOKI_HIP_BINARY(add, EK_ADD)    ENOKI_HIP_BINARY(sub, EK_SUB)    ENOKI_HIP_BINARY(mul, EK_MUL)    ENOKI_HIP_BINARY(div, EK_DIV)
    ENOKI_HIP_BINARY(mod, EK_MOD)    ENOKI_HIP_BINARY(mulhi, EK_MULHI) ENOKI_HIP_BINARY(min, EK_MIN)   ENOKI_HIP_BINARY(max, EK_MAX)
    ENOKI_HIP_BINARY(xor, EK_XOR)
    ENOKI_HIP_TERNARY(fmadd, EK_FMADD) ENOKI_HIP_TERNARY(fmsub, EK_FMSUB) ENOKI_HIP_TERNARY(fnmadd, EK_FNMADD) ENOKI_HIP_TERNARY(fnmsub, EK_FNMSUB)
    ENOKI_HIP_UNARY(neg, EK_NEG)     ENOKI_HIP_UNARY(abs, EK_ABS)     ENOKI_HIP_UNARY(not, EK_NOT)     ENOKI_HIP_UNARY(sqrt, EK_SQRT)
    ENOKI_HIP_UNARY(rcp, EK_RCP)     ENOKI_HIP_UNARY(rsqrt, EK_RSQRT) ENOKI_HIP_UNARY(floor, EK_FLOOR) ENOKI_HIP_UNARY(ceil, EK_CEIL)
    ENOKI_HIP_UNARY(round, EK_ROUND) ENOKI_HIP_UNARY(trunc, EK_TRUNC) ENOKI_HIP_UNARY(sin, EK_SIN)     ENOKI_HIP_UNARY(cos, EK_COS)
    ENOKI_HIP_UNARY(exp, EK_EXP)     ENOKI_HIP_UNARY(log, EK_LOG)     ENOKI_HIP_UNARY(popcnt, EK_POPCNT) ENOKI_HIP_UNARY(lzcnt, EK_LZCNT)
    ENOKI_HIP_UNARY(tzcnt, EK_TZCNT)
    ENOKI_HIP_COMPARE(eq, EK_EQ)     ENOKI_HIP_COMPARE(neq, EK_NEQ)   ENOKI_HIP_COMPARE(lt, EK_LT)     ENOKI_HIP_COMPARE(le, EK_LE)
    ENOKI_HIP_COMPARE(gt, EK_GT)     ENOKI_HIP_COMPARE(ge, EK_GE)
#undef ENOKI_HIP_UNARY
#undef ENOKI_HIP_BINARY
#undef ENOKI_HIP_TERNARY
#undef ENOKI_HIP_COMPARE

    std::pair<HIPArray, HIPArray> sincos_() const {                     // cuda.h:455-457
        HIPArray s = empty_(size()), c = empty_(size());
        ek_operand a = operand();
        hip_detail::check(ek_hip_sincos(Code, s.m_buf->ptr, c.m_buf->ptr, &a, size()), "sincos");
        return { s, c };
    }

    /// and_ / or_ / andnot_ with an operand of the same type are bit operations, with a mask they select (cuda.h:545-575)
    template <typename T> HIPArray and_(const HIPArray<T> &v) const {
        if constexpr (std::is_same_v<T, bool> && !std::is_same_v<Value, bool>) return select_(v, *this, HIPArray(Value(0)));
        else return bitop(EK_AND, v);
    }
    template <typename T> HIPArray or_(const HIPArray<T> &v) const {
        if constexpr (std::is_same_v<T, bool> && !std::is_same_v<Value, bool>)
            return select_(v, HIPArray(memcpy_cast<Value>(int_array_t<Value>(-1))), *this);
        else return bitop(EK_OR, v);
    }
    template <typename T> HIPArray andnot_(const HIPArray<T> &v) const {
        if constexpr (std::is_same_v<T, bool> && !std::is_same_v<Value, bool>) return select_(v, HIPArray(Value(0)), *this);
        else return bitop(EK_AND, v.not_());
    }

    template <size_t Imm> HIPArray sl_() const { return sl_(HIPArray((Value) Imm)); }
    template <size_t Imm> HIPArray sr_() const { return sr_(HIPArray((Value) Imm)); }
    HIPArray sl_(size_t k) const { return sl_(HIPArray((Value) k)); }
    HIPArray sr_(size_t k) const { return sr_(HIPArray((Value) k)); }
    HIPArray sl_(const HIPArray &v) const { return shift(EK_SL, v); }
    HIPArray sr_(const HIPArray &v) const { return shift(EK_SR, v); }     // arithmetic for signed types (cuda.h:499-523)

    template <typename T> T floor2int_() const { return T(floor_()); }
    template <typename T> T ceil2int_() const { return T(ceil_()); }

    static HIPArray select_(const MaskType &m, const HIPArray &t, const HIPArray &f) {      // cuda.h:632-639
        HIPArray r = empty_(broadcast(broadcast(m.size(), t.size()), f.size()));
        ek_operand om = m.operand(), ot = t.operand(), of = f.operand();
        hip_detail::check(ek_hip_select(Code, r.m_buf->ptr, &om, &ot, &of, r.size()), "select");
        return r;
    }

    // ---- initialisation (cuda.h:641-691) -----------------------------------------------------------------------------
    static HIPArray empty_(size_t size) { HIPArray r; r.allocate(size); return r; }
    static HIPArray zero_(size_t size) {
        HIPArray r = empty_(size);
        hip_detail::check(ek_hip_memset(r.m_buf->ptr, 0, size * sizeof(Value)), "zero");
        return r;
    }
    static HIPArray full_(const Value &value, size_t size) {
        HIPArray r = empty_(size);
        uint64_t bits = 0;
        memcpy(&bits, &value, sizeof(Value));
        hip_detail::check(ek_hip_fill(Code, r.m_buf->ptr, bits, size), "full");
        return r;
    }
    static HIPArray arange_(ssize_t start, ssize_t stop, ssize_t step) {
        size_t size = size_t((stop - start + step - (step > 0 ? 1 : -1)) / step);
        HIPArray r = empty_(size);
        hip_detail::check(ek_hip_arange(Code, r.m_buf->ptr, (int64_t) start, (int64_t) step, size), "arange");
        return r;
    }
    static HIPArray linspace_(Value min, Value max, size_t size) {
        HIPArray r = empty_(size);
        hip_detail::check(ek_hip_linspace(Code, r.m_buf->ptr, (double) min, (double) max, size), "linspace");
        return r;
    }
    static HIPArray map(void *ptr, size_t size, bool dealloc = false) {
        HIPArray r;
        r.m_buf = std::make_shared<hip_detail::Buffer>();
        r.m_buf->ptr = ptr; r.m_buf->size = size; r.m_buf->owned = dealloc;
        return r;
    }
    static HIPArray copy(const void *ptr, size_t size) {
        HIPArray r = empty_(size);
        hip_detail::check(ek_hip_memcpy_to_device(r.m_buf->ptr, ptr, size * sizeof(Value)), "copy");
        return r;
    }

    // ---- horizontal operations (cuda.h:693-794) ----------------------------------------------------------------------
#define ENOKI_HIP_REDUCE(name, code)                                                                                   \
    HIPArray name##_() const {                                                                                         \
        if (size() == 1) return *this;                                                                                 \
        HIPArray r = empty_(1);                                                                                        \
        hip_detail::check(ek_hip_reduce(code, Code, r.m_buf->ptr, m_buf ? m_buf->get() : nullptr, size()), #name);       \
        return r;                                                                                                      \
    }
    ENOKI_HIP_REDUCE(hsum, EK_HSUM) ENOKI_HIP_REDUCE(hprod, EK_HPROD) ENOKI_HIP_REDUCE(hmin, EK_HMIN) ENOKI_HIP_REDUCE(hmax, EK_HMAX)
#undef ENOKI_HIP_REDUCE
    bool all_() const { return mask_reduce(EK_ALL) != 0; }
    bool any_() const { return mask_reduce(EK_ANY) != 0; }
    size_t count_() const { return (size_t) mask_reduce(EK_COUNT); }
    HIPArray psum_() const {
        HIPArray r = empty_(size());
        hip_detail::check(ek_hip_psum(Code, r.m_buf->ptr, m_buf->get(), size()), "psum");
        return r;
    }
    HIPArray reverse_() const {
        HIPArray r = empty_(size());
        hip_detail::check(ek_hip_reverse(Code, r.m_buf->ptr, m_buf->get(), size()), "reverse");
        return r;
    }

    // ---- indexed memory operations (cuda.h:845-905); Stride == sizeof(Value) for arrays of Value -----------------------
    template <size_t Stride, typename Index_, typename Mask>
    static HIPArray gather_(const void *ptr, const Index_ &index, const Mask &mask) {
        static_assert(Stride == sizeof(Value), "HIPArray::gather_(): element stride expected");
        HIPArray r = empty_(broadcast(index.size(), mask.size()));
        int index_code = 0;
        ek_operand oi = index.index_operand(index_code), om = mask.operand();
        hip_detail::check(ek_hip_gather(Code, index_code, r.m_buf->ptr, ptr, &oi, &om, r.size()), "gather");
        return r;
    }
    template <size_t Stride, typename Index_, typename Mask>
    void scatter_(void *ptr, const Index_ &index, const Mask &mask) const {
        static_assert(Stride == sizeof(Value), "HIPArray::scatter_(): element stride expected");
        ek_operand ov = operand(), oi = index.operand(), om = mask.operand();
        hip_detail::check(ek_hip_scatter(Code, Index_::Code, ptr, &ov, &oi, &om, broadcast(broadcast(size(), index.size()), mask.size())), "scatter");
    }
    template <size_t Stride, typename Index_, typename Mask>
    void scatter_add_(void *ptr, const Index_ &index, const Mask &mask) const {
        static_assert(Stride == sizeof(Value), "HIPArray::scatter_add_(): element stride expected");
        int index_code = 0;
        ek_operand ov = operand(), oi = index.index_operand(index_code), om = mask.operand();
        const hip_detail::Operand &target = hip_detail::announced_operand();
        hip_detail::check(ek_hip_scatter_add(Code, index_code, ptr, target.ptr == ptr ? target.size : 0, &ov, &oi, &om,
                                             broadcast(broadcast(size(), index.size()), mask.size()), 0), "scatter_add");
    }

    /// Groups of equal instance pointers for vectorised method calls (cuda.h:814-843; array_router.h:671).  A compatibility
    /// implementation on the host; the production class sorts on the device (include/enoki/array_call.h, ek_hip_sort_pairs).
    template <typename T = Value, enable_if_t<std::is_pointer_v<T> || std::is_same_v<T, uintptr_t>> = 0>
    std::vector<std::pair<Value, HIPArray<uint32_t>>> partition_() const {
        std::vector<Value> host(size());
        if (!host.empty()) hip_detail::check(ek_hip_memcpy_to_host(host.data(), m_buf->get(), host.size() * sizeof(Value)), "partition");
        std::map<uintptr_t, std::vector<uint32_t>> groups;
        for (size_t i = 0; i < host.size(); ++i) groups[(uintptr_t) host[i]].push_back((uint32_t) i);
        std::vector<std::pair<Value, HIPArray<uint32_t>>> result;
        for (auto &g : groups) result.emplace_back((Value) g.first, HIPArray<uint32_t>::copy(g.second.data(), g.second.size()));
        return result;
    }
    auto operator->() const {
        using BaseType = std::decay_t<std::remove_pointer_t<Value>>;
        return call_support<BaseType, HIPArray>(*this);
    }

    // ---- storage (cuda.h:781-812, 930-949) ---------------------------------------------------------------------------
    HIPArray &eval() { return *this; }                 // eager backend: nothing is pending
    const HIPArray &eval() const { return *this; }
    HIPArray &managed() { return *this; }
    const HIPArray &managed() const { return *this; }
    Index index_() const { return m_buf ? hip_detail::Handles::get().park(m_buf) : 0; }   // see hip_detail::Handles
    static HIPArray from_index_(Index index) { HIPArray r; r.m_buf = hip_detail::Handles::get().find(index); return r; }
    size_t size() const { return m_buf ? m_buf->size : 0; }
    bool empty() const { return size() == 0; }
    const Value *data() const { return m_buf ? (const Value *) m_buf->get() : nullptr; }
    Value *data() {
        if (!m_buf) return nullptr;
        m_buf->version++;                                   // the caller may write: tags that describe the old contents expire
        m_buf->narrow_code = 0;
        m_buf->prod_kind = 0;
        return (Value *) m_buf->get();
    }
    void resize(size_t size) {
        if (size == this->size()) return;
        if (this->size() > 1) throw std::runtime_error("HIPArray::resize(): only size-1 arrays can be broadcast");
        HIPArray r = empty_(size);
        if (m_buf) {
            ek_operand a = operand();
            hip_detail::check(ek_hip_unary(EK_COPY, Code, r.m_buf->ptr, &a, size), "resize");
        }
        *this = r;
    }
    Value coeff(size_t i) const {
        Value result = Value(0);
        hip_detail::check(ek_hip_memcpy_to_host(&result, (const Value *) m_buf->get() + i, sizeof(Value)), "coeff");
        return result;
    }

    ek_operand operand() const { return ek_operand{ m_buf ? m_buf->get() : nullptr, 0, size() }; }

private:
    static size_t broadcast(size_t a, size_t b) {
        if (a == b || b == 1) return a;
        if (a == 1) return b;
        throw std::runtime_error("HIPArray: arrays of incompatible size (" + std::to_string(a) + " and " + std::to_string(b) + ")");
    }
    void allocate(size_t size) {
        m_buf = std::make_shared<hip_detail::Buffer>();
        m_buf->size = size;
        hip_detail::check(ek_hip_malloc((size ? size : 1) * sizeof(Value), &m_buf->ptr), "allocate");
    }
    template <typename T> HIPArray bitop(int code, const HIPArray<T> &v) const {
        static_assert(sizeof(T) == sizeof(Value));
        HIPArray r = empty_(broadcast(size(), v.size()));
        ek_operand a = operand(), b = v.operand();
        // bit operations on floating point data run on the integer type of the same width
        constexpr int IntCode = std::is_same_v<Value, bool> ? (int) EK_BOOL : sizeof(Value) == 4 ? (int) EK_U32 : (int) EK_U64;
        hip_detail::check(ek_hip_binary(code, IntCode, r.m_buf->ptr, &a, &b, r.size()), "bit operation");
        return r;
    }
    HIPArray shift(int code, const HIPArray &v) const {
        HIPArray r = empty_(broadcast(size(), v.size()));
        ek_operand a = operand(), b = v.operand();
        hip_detail::check(ek_hip_binary(code, Code, r.m_buf->ptr, &a, &b, r.size()), "shift");
        return r;
    }
    uint64_t mask_reduce(int code) const {
        static_assert(std::is_same_v<Value, bool>, "all / any / count need a mask array");
        uint64_t result = 0;
        hip_detail::check(ek_hip_mask_reduce(code, (const uint8_t *) (m_buf ? m_buf->get() : nullptr), size(), &result), "mask reduction");
        return result;
    }

    std::shared_ptr<hip_detail::Buffer> m_buf;
};

NAMESPACE_END(enoki)
