/*
    enoki/array_call.h -- device arrays of instance pointers and vectorised (virtual) method calls

    Same user-facing contract as the reference (include/enoki/array_call.h:17-283 + CUDAArray::partition_,
    cuda.h:815-842 / src/cuda/horiz.cu:35-122):

        struct Shape { virtual FloatC eval(const FloatC &x, const MaskC &active) const = 0; ... };
        ENOKI_CALL_SUPPORT_BEGIN(Shape)
        ENOKI_CALL_SUPPORT_METHOD(eval)
        ENOKI_CALL_SUPPORT_END(Shape)

        HIPArray<Shape *> shapes = ...;            // one instance pointer per lane (nullptr allowed)
        FloatC y = shapes->eval(x);                // one call per distinct instance on gathered arguments

    A call partitions the lanes by instance, gathers the arguments of each group through the group's
    permutation, calls the method once per instance with array arguments, and scatters the results
    back; lanes with a null pointer (or masked out) receive zero.

    MI355X-first difference: the reference sorts (pointer, lane) pairs with an 8-pass 64-bit CUB radix sort
    and run-length encodes them.  Pointer arrays in a renderer hold a handful of distinct instances, so
    partition_() here extracts one group per iteration -- masked u64 hmin (the smallest remaining pointer),
    compare, order-preserving compress of the lane indices: ~70 B per lane and instance, cheaper than the
    sort up to a few dozen instances -- and produces the same output: groups in ascending pointer order,
    lanes in ascending order within a group.  The partition is cached in the array (copies share it), like
    cuda.h:816-842.

    ENOKI_CALL_SUPPORT_GETTER (array_call.h:269-283) is provided for scalar data members: gathered out of instance memory
    on the device when the class declares ENOKI_PINNED_OPERATOR_NEW (pinned instances, like the reference), read on the
    host once per instance otherwise.
*/
#pragma once

#include <enoki/hip.h>

#include <memory>
#include <tuple>
#include <utility>

namespace enoki {

template <typename Class, typename Storage> struct call_support {
    call_support(const Storage &) { }
};

/// Device array of instance pointers, stored as 64-bit integers
template <typename Class_> struct HIPArray<Class_ *> : ArrayTag {
    static_assert(sizeof(void *) == sizeof(uint64_t), "64-bit pointers expected");

    using Value = Class_ *;
    using Scalar = Class_ *;
    using ArrayType = HIPArray;
    using MaskType = HIPArray<bool>;
    using UnderlyingType = HIPArray<uint64_t>;
    using Partition = std::vector<std::pair<Value, HIPArray<uint32_t>>>;
    template <typename T> using ReplaceValue = HIPArray<T>;
    template <typename T> using ReplaceScalar = HIPArray<T>;
    template <typename T> using ReplaceMaskValue = HIPArray<T>;

    static constexpr size_t Depth = 1;
    static constexpr size_t Rank = 2;
    static constexpr bool IsMask = false;
    static constexpr bool IsDiff = false;
    static constexpr bool IsDynamic = true;
    static constexpr bool IsDevice = true;
    static constexpr bool IsCUDA = false;
    static constexpr bool IsFloat = false;
    static constexpr bool IsInt = false;

    HIPArray() = default;
    HIPArray(Value p) : m_bits((uint64_t) (uintptr_t) p) { }
    HIPArray(std::nullptr_t) : m_bits(uint64_t(0)) { }
    explicit HIPArray(const UnderlyingType &bits) : m_bits(bits) { }

    /// Host pointer list -> device
    static HIPArray copy(const Value *ptrs, size_t size) {
        return HIPArray(UnderlyingType::copy((const uint64_t *) ptrs, size));
    }

    size_t size() const { return m_bits.size(); }
    size_t slices_() const { return m_bits.size(); }
    const UnderlyingType &bits() const { return m_bits; }
    Value coeff(size_t i) const { return (Value) (uintptr_t) m_bits.coeff(i); }
    Value operator[](size_t i) const { return coeff(i); }

    MaskType eq_(const HIPArray &o) const { return m_bits.eq_(o.m_bits); }
    MaskType neq_(const HIPArray &o) const { return m_bits.neq_(o.m_bits); }
    static HIPArray select_(const MaskType &m, const HIPArray &t, const HIPArray &f) {
        return HIPArray(UnderlyingType::select_(m, t.m_bits, f.m_bits));
    }
    template <bool IsPermute = false, typename Index>
    static HIPArray gather_array_(const HIPArray &source, const Index &index, const MaskType &mask) {
        return HIPArray(UnderlyingType::template gather_array_<IsPermute>(source.m_bits, index, mask));
    }

    /// Groups of lanes that share an instance: ascending pointer value, ascending lane index within a group
    /// (a null pointer forms a group of its own, which the call dispatcher skips).
    ///
    /// Like the reference (cuda_partition, horiz.cu:35-122: radix sort of (pointer, lane) pairs + run-length encoding), the
    /// cost does not depend on the number of distinct instances: the pointers are mapped to dense 32-bit keys
    /// ((p - lowest) / 8 + 1, 0 for null -- as many key bits as the instances' address span needs), (key, lane) pairs go
    /// through ONE stable LSD radix sort (ek_hip_sort_pairs), run starts come from a compare with the left neighbour + an
    /// order-preserving compress.  Groups are views into the sorted lane array.  Pointer sets that do not fit 32-bit keys
    /// (spans >= 32 GiB, pointers that are not 8-byte aligned) take the instance-by-instance extraction below.
    const Partition &partition_() const {
        if (!m_partition) {
            auto result = std::make_shared<Partition>();
            size_t n = size();
            using UInt32 = HIPArray<uint32_t>;
            if (n == 1) {
                result->emplace_back(coeff(0), UInt32(0u));
            } else if (n > 1 && !partition_sorted_(*result)) {
                const uint64_t none = ~uint64_t(0);
                MaskType remaining = MaskType::full_(true, n);
                UInt32 lane = UInt32::arange_(0, (ptrdiff_t) n, 1);
                while (true) {
                    uint64_t value = UnderlyingType::select_(remaining, m_bits, UnderlyingType(none)).hmin_().coeff(0);
                    if (value == none)
                        break;
                    MaskType group = m_bits.eq_(UnderlyingType(value));
                    result->emplace_back((Value) (uintptr_t) value, lane.compress_(group));
                    remaining = remaining.and_(group.not_());
                }
            }
            m_partition = std::move(result);
        }
        return *m_partition;
    }

    auto operator->() const {
        using BaseType = std::decay_t<Class_>;
        return call_support<BaseType, HIPArray>(*this);
    }

private:
    /// sort-based partition (see partition_()); false when the pointers do not map to 32-bit keys
    bool partition_sorted_(Partition &result) const {
        using UInt32 = HIPArray<uint32_t>;
        using UInt64 = UnderlyingType;
        const size_t n = size();
        if (n >= ((size_t) 1 << 32)) return false;
        const UInt64 zero(uint64_t(0));
        MaskType is_null = m_bits.eq_(zero);
        const uint64_t lowest = UInt64::select_(is_null, UInt64(~uint64_t(0)), m_bits).hmin_().coeff(0);
        if (lowest == ~uint64_t(0)) {                         // every lane is null
            result.emplace_back((Value) nullptr, UInt32::arange_(0, (ptrdiff_t) n, 1));
            return true;
        }
        const uint64_t highest = m_bits.hmax_().coeff(0);
        const uint64_t misaligned = m_bits.or_(UInt64(lowest)).and_(UInt64(uint64_t(7))).hmax_().coeff(0);
        const uint64_t span = ((highest - lowest) >> 3) + 2;       // keys 1 .. span - 1, 0 = null
        if (misaligned || span > (uint64_t(1) << 32)) return false;
        int key_bits = 1;
        while ((uint64_t(1) << key_bits) < span) ++key_bits;
        UInt32 keys = UInt32(UInt64::select_(is_null, zero, m_bits.sub_(UInt64(lowest)).sr_(UInt64(uint64_t(3))).add_(UInt64(uint64_t(1)))));
        UInt32 sorted = UInt32::empty_(n), lanes = UInt32::empty_(n);
        detail::hip_check(ek_hip_sort_pairs(key_bits, keys.data(), n, sorted.data(), lanes.data()), "partition_");
        // run starts: entries whose key differs from the left neighbour's (the first entry always starts a run)
        UInt32 pos = UInt32::arange_(0, (ptrdiff_t) n, 1);
        UInt32 left = UInt32::template gather_<sizeof(uint32_t)>(sorted.data(), pos.sub_(UInt32(1u)), pos.neq_(UInt32(0u)));
        MaskType starts_here = sorted.neq_(left).or_(pos.eq_(UInt32(0u)));
        std::vector<uint32_t> starts = pos.compress_(starts_here).to_host();
        std::vector<uint32_t> start_keys = sorted.compress_(starts_here).to_host();
        m_partition_lanes = lanes;                              // the groups below are views into this array
        for (size_t g = 0; g < starts.size(); ++g) {
            const size_t begin = starts[g], end = g + 1 < starts.size() ? starts[g + 1] : n;
            const uint64_t p = start_keys[g] == 0 ? 0 : lowest + ((uint64_t) (start_keys[g] - 1) << 3);
            result.emplace_back((Value) (uintptr_t) p, UInt32::view_(m_partition_lanes, begin, end - begin));   // shares ownership
        }
        return true;
    }

    UnderlyingType m_bits;
    mutable std::shared_ptr<Partition> m_partition;
    mutable HIPArray<uint32_t> m_partition_lanes;      // storage behind the sort-based partition's group views
};

template <typename T, enable_if_t<is_array_v<T>> = 0> inline decltype(auto) partition(const T &a) { return a.partition_(); }

namespace detail {
    /// Does `Expr<Args...>` name a valid expression?
    template <typename, template <typename...> typename Expr, typename... Args> struct is_valid_call : std::false_type { };
    template <template <typename...> typename Expr, typename... Args>
    struct is_valid_call<std::void_t<Expr<Args...>>, Expr, Args...> : std::true_type { };
    template <template <typename...> typename Expr, typename... Args>
    constexpr bool is_callable_v = is_valid_call<void, Expr, Args...>::value;

    /// A method that returns a plain scalar yields one value per lane
    template <typename Result, typename = int> struct vectorize_result { using type = Result; };
    template <typename Result> struct vectorize_result<Result, enable_if_t<std::is_arithmetic_v<Result>>> {
        using type = HIPArray<Result>;
    };

    /// Arguments that are arrays travel through the group's permutation; scalars and size-1 arrays pass through
    template <typename T, typename Perm> inline auto gather_argument(const T &v, const Perm &perm) {
        if constexpr (is_array_v<T> || is_struct_v<T>) {      // arrays, nested arrays and ENOKI_STRUCT types
            if (slices(v) <= 1)
                return T(v);
            return gather<T, 0, true, true>(v, perm);
        } else {
            return T(v);
        }
    }

    template <typename... Ts> constexpr bool last_is_mask() {
        if constexpr (sizeof...(Ts) == 0) {
            return false;
        } else {
            using Last = std::tuple_element_t<sizeof...(Ts) - 1, std::tuple<std::decay_t<Ts>...>>;
            return is_mask_v<Last>;
        }
    }

    template <typename Storage_> struct call_support_base {
        using Storage = Storage_;
        using InstancePtr = typename Storage_::Value;
        using Mask = typename Storage_::MaskType;

        call_support_base(const Storage &storage) : self(storage) { }
        const Storage &self;

        /// func(instance, mask, args...) once per distinct non-null instance (array_call.h:124-193, device branch)
        template <typename Func, typename InputMask, typename Tuple, size_t... Indices>
        auto dispatch(Func func, const InputMask &mask_, const Tuple &tuple, std::index_sequence<Indices...>) const {
            Mask mask = Mask(mask_).and_(self.neq_(Storage(nullptr)));
            using FuncResult = decltype(func(std::declval<InstancePtr>(), mask, std::get<Indices>(tuple)...));
            const auto &groups = self.partition_();
            const bool single = groups.size() == 1 && groups[0].first != nullptr;

            if constexpr (!std::is_void_v<FuncResult>) {
                using Result = typename vectorize_result<FuncResult>::type;
                if (single)      // every lane holds the same instance: no permutation needed
                    return Result(func(groups[0].first, Mask(true), std::get<Indices>(tuple)...));
                Result result = zero<Result>(self.size());
                for (const auto &[instance, permutation] : groups) {
                    if (instance == nullptr)
                        continue;
                    Result part = func(instance, gather_argument(mask, permutation),
                                       gather_argument(std::get<Indices>(tuple), permutation)...);
                    scatter<0, true, true>(result, part, permutation);
                }
                return result;
            } else {
                if (single) {
                    func(groups[0].first, Mask(true), std::get<Indices>(tuple)...);
                    return;
                }
                for (const auto &[instance, permutation] : groups) {
                    if (instance == nullptr)
                        continue;
                    func(instance, gather_argument(mask, permutation),
                         gather_argument(std::get<Indices>(tuple), permutation)...);
                }
            }
        }
    };
}

#define ENOKI_CALL_SUPPORT_FRIEND()                                                               \
    template <typename, typename> friend struct enoki::call_support;

#define ENOKI_CALL_SUPPORT_BEGIN(Class_)                                                          \
    namespace enoki {                                                                             \
    template <typename Storage> struct call_support<Class_, Storage> : detail::call_support_base<Storage> { \
        using Base = detail::call_support_base<Storage>;                                          \
        using Base::Base;                                                                         \
        using typename Base::Mask;                                                                \
        using typename Base::InstancePtr;                                                         \
        using Class = Class_;                                                                     \
        using Base::self;                                                                         \
        auto operator->() { return this; }

/// `ptrs->name(args..., [mask])`: the method receives the lane mask as a trailing argument when it accepts one
#define ENOKI_CALL_SUPPORT_METHOD(name)                                                           \
    private:                                                                                      \
        template <typename... Args>                                                               \
        using name##_result_t = decltype(std::declval<InstancePtr>()->name(std::declval<Args>()...)); \
    public:                                                                                       \
        template <typename... Args> auto name(Args &&... args) const {                            \
            auto invoke = [](InstancePtr instance, const Mask &active, const auto &... a) {       \
                (void) active;                                                                    \
                if constexpr (detail::is_callable_v<name##_result_t, decltype(a)..., Mask>)       \
                    return instance->name(a..., active);                                          \
                else                                                                              \
                    return instance->name(a...);                                                  \
            };                                                                                    \
            auto packed = std::tie(args...);                                                      \
            if constexpr (detail::last_is_mask<Args...>())                                        \
                return Base::dispatch(invoke, std::get<sizeof...(Args) - 1>(packed), packed,      \
                                      std::make_index_sequence<sizeof...(Args) - 1>());           \
            else                                                                                  \
                return Base::dispatch(invoke, true, packed, std::make_index_sequence<sizeof...(Args)>()); \
        }

/// Instances that the GPU can read: classes that declare ENOKI_PINNED_OPERATOR_NEW(Type) allocate themselves in pinned host
/// memory (ek_hip_host_malloc; the reference's macro of the same name uses cuda_host_malloc, array_macro.h:361-395) and are
/// marked, so that ENOKI_CALL_SUPPORT_GETTER reads their data members ON THE DEVICE.  As in the reference, instances of such
/// a class must then live on the heap (`new`): a stack object is ordinary host memory.
#define ENOKI_PINNED_OPERATOR_NEW(Type)                                                           \
    static constexpr bool enoki_pinned_instances_ = true;                                         \
    void *operator new(size_t size) { return enoki::detail::pinned_new(size); }                   \
    void *operator new(size_t size, std::align_val_t) { return enoki::detail::pinned_new(size); } \
    void *operator new[](size_t size) { return enoki::detail::pinned_new(size); }                 \
    void *operator new[](size_t size, std::align_val_t) { return enoki::detail::pinned_new(size); } \
    void operator delete(void *ptr) { ek_hip_host_free(ptr); }                                    \
    void operator delete(void *ptr, std::align_val_t) { ek_hip_host_free(ptr); }                  \
    void operator delete[](void *ptr) { ek_hip_host_free(ptr); }                                  \
    void operator delete[](void *ptr, std::align_val_t) { ek_hip_host_free(ptr); }

namespace detail {
    inline void *pinned_new(size_t size) {
        void *p = nullptr;
        if (ek_hip_host_malloc(size, &p) != EK_OK) throw std::bad_alloc();
        return p;
    }
    template <typename T, typename = void> struct is_pinned_class : std::false_type { };
    template <typename T> struct is_pinned_class<T, std::enable_if_t<T::enoki_pinned_instances_>> : std::true_type { };
}

/// `ptrs->name()`: per-lane value of a scalar data member of the instances (array_call.h:269-283).  Like the reference, the
/// field is gathered straight out of instance memory ON THE DEVICE (ek_hip_gather_address: one kernel, whatever the number of
/// instances) when the class allocates its instances where the GPU can read them (ENOKI_PINNED_OPERATOR_NEW above); instances
/// in ordinary host memory are read on the host once per distinct instance and scattered to that instance's lanes.
/// Null pointers and masked lanes give 0 either way.
#define ENOKI_CALL_SUPPORT_GETTER_TYPE(name, field, type)                                         \
    HIPArray<type> name(Mask mask = Mask(true)) const {                                           \
        using Return = HIPArray<type>;                                                            \
        using FieldType_ = std::decay_t<decltype(std::declval<Class>().field)>;                   \
        if constexpr (enoki::detail::is_pinned_class<Class>::value && std::is_arithmetic_v<FieldType_> && \
                      !std::is_same_v<FieldType_, bool> &&                                        \
                      (sizeof(FieldType_) == 1 || sizeof(FieldType_) == 4 || sizeof(FieldType_) == 8)) { /* what the kernel reads */ \
            const ptrdiff_t offset_ = (ptrdiff_t) (uintptr_t) &(((Class *) nullptr)->field);      \
            return Return(HIPArray<FieldType_>::gather_address_(self.bits(), offset_, mask));     \
        }                                                                                         \
        mask = mask.and_(self.neq_(Storage(nullptr)));                                            \
        const auto &groups = self.partition_();                                                   \
        if (groups.size() == 1 && groups[0].first != nullptr)                                     \
            return Return::select_(mask, Return((type) groups[0].first->field), Return(type(0)));  \
        Return result = zero<Return>(self.size());                                                \
        for (const auto &[instance, permutation] : groups) {                                      \
            if (instance == nullptr)                                                              \
                continue;                                                                         \
            scatter<0, true, true>(result, Return((type) instance->field), permutation,           \
                                   detail::gather_argument(mask, permutation));                   \
        }                                                                                         \
        return result;                                                                            \
    }

#define ENOKI_CALL_SUPPORT_GETTER(name, field)                                                    \
    ENOKI_CALL_SUPPORT_GETTER_TYPE(name, field, std::decay_t<decltype(std::declval<Class>().field)>)

#define ENOKI_CALL_SUPPORT_END(Class_)                                                            \
        };                                                                                        \
    }

} // namespace enoki
