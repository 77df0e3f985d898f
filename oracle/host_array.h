/*
 * oracle/host_array.h -- TEST INFRASTRUCTURE ONLY.
 *
 * HostArray<T>: a CPU array type with the same member concept as HIPArray<T>, implemented on top of
 * the C oracle (oracle/enoki_oracle.c, i.e. the restated AVX2 DynamicArray<Packet<T,8>> semantics).
 * Instantiating the product's generic DiffArray / Tape templates (include/enoki/autodiff.h,
 * enoki_amd/src/autodiff_impl.h) over this type lets the CPU-only test suite check the HOST logic of
 * the tape (graph bookkeeping, sweep order, refcounts, specials) bit-for-bit against the reference
 * build (oracle/_ref) without a GPU.  It is never linked into the product.
 */
#pragma once

#include <enoki/array.h>

#include <memory>
#include <vector>

extern "C" {
int orc_unary(int type, const char *op, const void *a, void *out, size_t n);
int orc_sincos(int type, const void *a, void *s, void *c, size_t n);
int orc_binary(int type, const char *op, const void *a, const void *b, void *out, size_t n);
int orc_ternary(int type, const char *op, const void *a, const void *b, const void *c, void *out, size_t n);
int orc_compare(int type, const char *op, const void *a, const void *b, uint8_t *out, size_t n);
int orc_select(int type, const uint8_t *m, const void *t, const void *f, void *out, size_t n);
int orc_cast(int src, int dst, const void *a, void *out, size_t n);
int orc_gather(int type, int itype, const void *base, size_t src_size, const void *idx, const uint8_t *mask,
               void *out, size_t n);
int orc_scatter(int type, int itype, int add, void *base, const void *val, const void *idx, const uint8_t *mask,
                size_t n);
int orc_reduce(int type, const char *op, const void *a, void *out, size_t n);
int orc_mask_reduce(const char *op, const uint8_t *m, uint64_t *out, size_t n);
int orc_psum_f32(const float *a, float *out, size_t n);
}

namespace enoki {

namespace detail {
    template <typename T> struct orc_type;
    template <> struct orc_type<bool>     { static constexpr int value = 0; };
    template <> struct orc_type<int32_t>  { static constexpr int value = 1; };
    template <> struct orc_type<uint32_t> { static constexpr int value = 2; };
    template <> struct orc_type<int64_t>  { static constexpr int value = 3; };
    template <> struct orc_type<uint64_t> { static constexpr int value = 4; };
    template <> struct orc_type<float>    { static constexpr int value = 5; };
    template <> struct orc_type<double>   { static constexpr int value = 6; };
}

template <typename Value_> struct HostArray : ArrayTag {
    template <typename T> friend struct HostArray;
    using Value = Value_;
    using Scalar = Value_;
    using Store = std::conditional_t<std::is_same_v<Value, bool>, uint8_t, Value>;
    using MaskType = HostArray<bool>;
    template <typename T> using ReplaceScalar = HostArray<T>;
    template <typename T> using ReplaceValue = HostArray<T>;
    template <typename T> using ReplaceMaskValue = HostArray<T>;
    static constexpr int Type = detail::orc_type<Value>::value;
    static constexpr size_t Depth = 1, Rank = 2;
    static constexpr bool IsMask = std::is_same_v<Value, bool>, IsDiff = false, IsDynamic = true, IsDevice = false;

    HostArray() = default;
    HostArray(Value v) : m_data(std::make_shared<std::vector<Store>>(1, (Store) v)) { }
    template <typename T, enable_if_t<std::is_arithmetic_v<T> && !std::is_same_v<T, Value>> = 0>
    HostArray(T v) : HostArray((Value) v) { }
    template <typename T, enable_if_t<!std::is_same_v<T, Value>> = 0> HostArray(const HostArray<T> &v) {
        if (!v.m_data) return;
        alloc(v.size());
        check(orc_cast(HostArray<T>::Type, Type, v.raw(), raw_mut(), size()), "cast");
    }
    template <typename T> HostArray(const HostArray<T> &v, detail::reinterpret_flag) {
        static_assert(sizeof(T) == sizeof(Value));
        if (!v.m_data) return;
        alloc(v.size());
        memcpy(raw_mut(), v.raw(), size() * sizeof(Store));
    }

    static HostArray copy(const void *ptr, size_t n) {
        HostArray r; r.alloc(n);
        if (n) memcpy(r.raw_mut(), ptr, n * sizeof(Store));
        return r;
    }
    static HostArray empty_(size_t n) { HostArray r; r.alloc(n); return r; }
    static HostArray zero_(size_t n) { HostArray r; r.alloc(n); return r; }
    static HostArray full_(const Value &v, size_t n) { HostArray r; r.m_data = std::make_shared<std::vector<Store>>(n, (Store) v); return r; }

    size_t size() const { return m_data ? m_data->size() : 0; }
    size_t slices_() const { return size(); }
    bool empty() const { return size() == 0; }
    const Store *raw() const { return m_data ? m_data->data() : nullptr; }
    const Value *data() const { return (const Value *) raw(); }
    Value *data() { unshare(); return (Value *) raw_mut(); }
    Value coeff(size_t i) const { return (Value) (*m_data)[i]; }
    void resize(size_t n) { set_slices_(n); }
    void set_slices_(size_t n) {
        if (size() == n) return;
        if (size() == 0) { alloc(n); return; }
        if (size() != 1) throw std::runtime_error("HostArray::resize(): only arrays of size 0 or 1 can be resized");
        Store v = (*m_data)[0];
        m_data = std::make_shared<std::vector<Store>>(n, v);
    }
    HostArray &eval() { return *this; }
    const HostArray &eval() const { return *this; }
    HostArray &managed() { return *this; }

#define HOST_UNARY(name, op) HostArray name##_() const { return unary(op); }
    HOST_UNARY(neg, "neg") HOST_UNARY(abs, "abs") HOST_UNARY(sqrt, "sqrt") HOST_UNARY(rcp, "rcp")
    HOST_UNARY(rsqrt, "rsqrt") HOST_UNARY(floor, "floor") HOST_UNARY(ceil, "ceil") HOST_UNARY(round, "round")
    HOST_UNARY(trunc, "trunc") HOST_UNARY(sin, "sin") HOST_UNARY(cos, "cos") HOST_UNARY(exp, "exp")
    HOST_UNARY(log, "log") HOST_UNARY(sign, "sign") HOST_UNARY(popcnt, "popcnt") HOST_UNARY(lzcnt, "lzcnt")
    HOST_UNARY(tzcnt, "tzcnt") HOST_UNARY(tan, "tan") HOST_UNARY(cot, "cot") HOST_UNARY(asin, "asin")
    HOST_UNARY(acos, "acos") HOST_UNARY(atan, "atan") HOST_UNARY(sinh, "sinh") HOST_UNARY(cosh, "cosh")
    HOST_UNARY(tanh, "tanh") HOST_UNARY(asinh, "asinh") HOST_UNARY(acosh, "acosh") HOST_UNARY(atanh, "atanh")
    HOST_UNARY(cbrt, "cbrt")
#undef HOST_UNARY
    std::pair<HostArray, HostArray> sincosh_() const { return { sinh_(), cosh_() }; }
    HostArray not_() const {
        if constexpr (IsMask) { HostArray r = empty_(size()); for (size_t i = 0; i < size(); ++i) (*r.m_data)[i] = !(*m_data)[i]; return r; }
        else return unary("not");
    }
#define HOST_BINARY(name, op) HostArray name##_(const HostArray &b) const { return binary(op, b); }
    HOST_BINARY(add, "add") HOST_BINARY(sub, "sub") HOST_BINARY(mul, "mul") HOST_BINARY(div, "div")
    HOST_BINARY(mod, "mod") HOST_BINARY(min, "min") HOST_BINARY(max, "max") HOST_BINARY(mulhi, "mulhi")
    HOST_BINARY(xor, "xor") HOST_BINARY(sl, "sl") HOST_BINARY(sr, "sr") HOST_BINARY(atan2, "atan2")
    HOST_BINARY(ldexp, "ldexp")
#undef HOST_BINARY
    HostArray and_(const HostArray &b) const { return bitop(b, 0); }
    HostArray or_(const HostArray &b) const { return bitop(b, 1); }
    template <typename T = Value, enable_if_t<!std::is_same_v<T, bool>> = 0>
    HostArray and_(const MaskType &m) const { return select_(m, *this, HostArray(Value(0))); }
    template <typename T = Value, enable_if_t<!std::is_same_v<T, bool>> = 0>
    HostArray or_(const MaskType &m) const {
        Value ones; memset(&ones, 0xff, sizeof(Value));
        return select_(m, HostArray(ones), *this);
    }
#define HOST_TERNARY(name, op) HostArray name##_(const HostArray &b, const HostArray &c) const { return ternary(op, b, c); }
    HOST_TERNARY(fmadd, "fmadd") HOST_TERNARY(fmsub, "fmsub") HOST_TERNARY(fnmadd, "fnmadd") HOST_TERNARY(fnmsub, "fnmsub")
#undef HOST_TERNARY
#define HOST_COMPARE(name, op) MaskType name##_(const HostArray &b) const { return compare(op, b); }
    HOST_COMPARE(eq, "eq") HOST_COMPARE(neq, "neq") HOST_COMPARE(lt, "lt") HOST_COMPARE(le, "le")
    HOST_COMPARE(gt, "gt") HOST_COMPARE(ge, "ge")
#undef HOST_COMPARE

    std::pair<HostArray, HostArray> sincos_() const {
        HostArray s = empty_(size()), c = empty_(size());
        check(orc_sincos(Type, raw(), s.raw_mut(), c.raw_mut(), size()), "sincos");
        return { s, c };
    }

    static HostArray select_(const MaskType &m, const HostArray &t, const HostArray &f) {
        size_t n = bsize(bsize(m.size(), t.size()), f.size());
        MaskType mm = m.expanded(n); HostArray tt = t.expanded(n), ff = f.expanded(n), r = empty_(n);
        check(orc_select(Type, mm.raw(), tt.raw(), ff.raw(), r.raw_mut(), n), "select");
        return r;
    }

    template <bool IsPermute, typename Index>
    static HostArray gather_array_(const HostArray &source, const Index &index, const MaskType &mask) {
        if (source.size() <= 1) return source & mask;
        size_t n = bsize(index.size(), mask.size());
        Index ii = index.expanded(n); MaskType mm = mask.expanded(n); HostArray r = empty_(n);
        check(orc_gather(Type, Index::Type, source.raw(), source.size(), ii.raw(), mm.raw(), r.raw_mut(), n), "gather");
        return r;
    }
    template <bool IsPermute, typename Index>
    static void scatter_array_(HostArray &target, const HostArray &value, const Index &index, const MaskType &mask) {
        scatter_impl(target, value, index, mask, 0);
    }
    template <bool IsPermute, typename Index>
    static void scatter_add_array_(HostArray &target, const HostArray &value, const Index &index, const MaskType &mask) {
        scatter_impl(target, value, index, mask, 1);
    }

    HostArray hsum_() const { return reduce("hsum"); }
    HostArray hprod_() const { return reduce("hprod"); }
    HostArray hmin_() const { return reduce("hmin"); }
    HostArray hmax_() const { return reduce("hmax"); }
    bool all_() const { return mreduce("all") != 0; }
    bool any_() const { return mreduce("any") != 0; }
    size_t count_() const { return (size_t) mreduce("count"); }
    HostArray reverse_() const {
        HostArray r = empty_(size());
        for (size_t i = 0; i < size(); ++i) (*r.m_data)[i] = (*m_data)[size() - 1 - i];
        return r;
    }
    HostArray psum_() const {
        static_assert(std::is_same_v<Value, float> || !std::is_same_v<Value, Value>, "psum: f32 only in the oracle");
        HostArray r = empty_(size());
        orc_psum_f32((const float *) raw(), (float *) r.raw_mut(), size());
        return r;
    }

    HostArray expanded(size_t n) const {
        if (size() == n) return *this;
        HostArray r = *this; r.set_slices_(n); return r;
    }

private:
    static void check(int rc, const char *what) {
        if (rc) throw std::runtime_error(std::string("HostArray: oracle does not implement ") + what);
    }
    static size_t bsize(size_t a, size_t b) {
        if (a == b || b == 1) return a;
        if (a == 1) return b;
        throw std::runtime_error("HostArray: arrays of incompatible size");
    }
    void alloc(size_t n) { m_data = std::make_shared<std::vector<Store>>(n, Store(0)); }
    void unshare() { if (m_data && m_data.use_count() > 1) m_data = std::make_shared<std::vector<Store>>(*m_data); }
    Store *raw_mut() { return m_data ? m_data->data() : nullptr; }

    HostArray unary(const char *op) const {
        HostArray r = empty_(size());
        check(orc_unary(Type, op, raw(), r.raw_mut(), size()), op);
        return r;
    }
    HostArray binary(const char *op, const HostArray &b) const {
        size_t n = bsize(size(), b.size());
        HostArray aa = expanded(n), bb = b.expanded(n), r = empty_(n);
        check(orc_binary(Type, op, aa.raw(), bb.raw(), r.raw_mut(), n), op);
        return r;
    }
    HostArray bitop(const HostArray &b, int which) const {
        size_t n = bsize(size(), b.size());
        HostArray aa = expanded(n), bb = b.expanded(n), r = empty_(n);
        const uint8_t *pa = (const uint8_t *) aa.raw(), *pb = (const uint8_t *) bb.raw();
        uint8_t *pr = (uint8_t *) r.raw_mut();
        for (size_t i = 0; i < n * sizeof(Store); ++i) pr[i] = which ? (pa[i] | pb[i]) : (pa[i] & pb[i]);
        return r;
    }
    HostArray ternary(const char *op, const HostArray &b, const HostArray &c) const {
        size_t n = bsize(bsize(size(), b.size()), c.size());
        HostArray aa = expanded(n), bb = b.expanded(n), cc = c.expanded(n), r = empty_(n);
        check(orc_ternary(Type, op, aa.raw(), bb.raw(), cc.raw(), r.raw_mut(), n), op);
        return r;
    }
    MaskType compare(const char *op, const HostArray &b) const {
        size_t n = bsize(size(), b.size());
        HostArray aa = expanded(n), bb = b.expanded(n);
        MaskType r = MaskType::empty_(n);
        check(orc_compare(Type, op, aa.raw(), bb.raw(), (uint8_t *) r.data(), n), op);
        return r;
    }
    HostArray reduce(const char *op) const {
        if (size() == 1) return *this;
        HostArray r = empty_(1);
        check(orc_reduce(Type, op, raw(), r.raw_mut(), size()), op);
        return r;
    }
    uint64_t mreduce(const char *op) const {
        uint64_t out = 0;
        orc_mask_reduce(op, (const uint8_t *) raw(), &out, size());
        return out;
    }
    template <typename Index>
    static void scatter_impl(HostArray &target, const HostArray &value, const Index &index, const MaskType &mask, int add) {
        size_t n = bsize(bsize(value.size(), index.size()), mask.size());
        HostArray vv = value.expanded(n); Index ii = index.expanded(n); MaskType mm = mask.expanded(n);
        target.unshare();
        check(orc_scatter(Type, Index::Type, add, target.raw_mut(), vv.raw(), ii.raw(), mm.raw(), n), "scatter");
    }

    std::shared_ptr<std::vector<Store>> m_data;
};

} // namespace enoki
