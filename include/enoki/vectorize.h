/*
    enoki/vectorize.h -- compile-time kernel fusion for the HIP backend: enoki::vectorize(f, args...)

    The reference's answer to "an eager array library launches one kernel per operation" on the CPU is structured
    vectorization: the user writes a function template over a `Value` type, and `vectorize()` (dynamic.h:1025-1074) runs
    it packet by packet over dynamic arrays, so that all intermediates stay in registers.  On its GPU backend the JIT
    obtains the same effect at run time (jit.cu:1066-1217).  This header gives the HIP backend the compile-time variant --
    no JIT, no code generation at run time: `vectorize(f, args...)` instantiates `f` on ONE-ELEMENT packets
    (Array<float, 1>, Array<Array<float, 1>, 3>, ENOKI_STRUCT types thereof) inside ONE __global__ kernel,

        out[i] = f(packet(args, i)...)        for every slice i, one slice per lane,

    so a chain like tests/sphere.cpp's make_rays -> intersect_rays -> shade_hits costs the loads of its inputs and the
    stores of its outputs instead of ~40 kernel launches with 300 B of intermediate traffic per element.

    Contract (same as the reference):
      * arguments that are dynamic (HIPArray, Array<HIPArray, N>, ENOKI_STRUCT types with dynamic fields) are sliced;
        size-1 arrays broadcast; everything else is passed to every invocation by value;
      * the result type is make_dynamic_t of what `f` returns (arrays, nested arrays, ENOKI_STRUCT types, or void);
      * non-const lvalue arguments are written back after the call (the reference hands out references into the storage),
        so `f` may update its arguments in place; pass const references to avoid the write-back traffic;
      * arguments of incompatible length throw std::runtime_error("vectorize(): vector arguments have incompatible lengths").

    Requirements: the translation unit is compiled by hipcc for gfx950 with -ffp-contract=off (only explicit fmadd()
    calls fuse, like everywhere else in this backend), this header is the FIRST enoki header it includes (it makes the
    array vocabulary callable from device code), and the user's own templates are bracketed by
    ENOKI_DEVICE_CODE_BEGIN / ENOKI_DEVICE_CODE_END.  Arithmetic inside `f` is IEEE (correctly rounded + - * / sqrt,
    fma); sin, cos, sincos, tan, exp, log, asin, acos, atan, atan2, sinh, cosh, tanh, pow, cbrt are the device algorithms
    of the stand-alone kernels (enoki/device/ek_math.h): a fused kernel returns the same bits as the op-by-op program.
*/
#pragma once

#if !defined(__HIP__)
#  error "enoki/vectorize.h instantiates user code inside a __global__ kernel: compile this translation unit with hipcc (-x hip --offload-arch=gfx950 -ffp-contract=off)"
#endif

/// Functions declared between these two markers can be called from vectorize() kernels
#define ENOKI_DEVICE_CODE_BEGIN _Pragma("clang force_cuda_host_device begin")
#define ENOKI_DEVICE_CODE_END   _Pragma("clang force_cuda_host_device end")

// system headers and the C ABI first: they must NOT be seen (for the first time) inside the device-code brackets
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#include <enoki_hip.h>
// the device algorithms of the stand-alone kernels: inside a fused kernel sin / cos / exp / log / tan / asin / ... of a
// packet are the SAME functions, hence the same bits, as the op-by-op path (array.h routes its scalar functions here)
#include <enoki/device/ek_math.h>
#define ENOKI_HIP_DEVICE_MATH 1

ENOKI_DEVICE_CODE_BEGIN
#include <enoki/array.h>
#include <enoki/hip.h>
ENOKI_DEVICE_CODE_END

namespace enoki {

ENOKI_DEVICE_CODE_BEGIN

namespace detail {
    // ---- one-element packets <-> dynamic types --------------------------------------------------------------------
    template <typename T> struct is_hip_array : std::false_type { };
    template <typename T> struct is_hip_array<HIPArray<T>> : std::true_type { };

    template <typename T> struct is_lane : std::false_type { };                         // Array<arithmetic, 1>
    template <typename T> struct is_lane<Array<T, 1>> : std::bool_constant<std::is_arithmetic_v<T>> { };

    /// Is `T` (or, for structures, any field of it) stored in device arrays?
    template <typename T, typename = int> struct has_dynamic_storage : std::bool_constant<is_array_v<T> && is_dynamic_v<T>> { };

    template <typename D, typename = int> struct packet_of { using type = D; };          // scalars pass through
    template <typename T> struct packet_of<HIPArray<T>> { using type = Array<T, 1>; };
    template <typename V, size_t N> struct packet_of<Array<V, N>> { using type = Array<typename packet_of<V>::type, N>; };
    template <template <typename...> class S, typename... A>
    struct packet_of<S<A...>, enable_if_t<is_struct_v<S<A...>>>> { using type = S<typename packet_of<A>::type...>; };

    template <typename T, size_t N>                                      // std::array with enoki/stl.h included
    struct packet_of<std::array<T, N>, enable_if_t<is_struct_v<std::array<T, N>>>> { using type = std::array<typename packet_of<T>::type, N>; };

    template <typename P, typename = int> struct dynamic_of { using type = P; };
    template <> struct dynamic_of<void> { using type = void; };
    template <typename T> struct dynamic_of<T, enable_if_t<std::is_arithmetic_v<T>>> { using type = HIPArray<T>; };
    template <typename V, size_t N> struct dynamic_of<Array<V, N>, int> {
        using type = std::conditional_t<is_lane<Array<V, N>>::value, HIPArray<scalar_t<V>>,
                                        Array<typename dynamic_of<V>::type, N>>;
    };
    template <template <typename...> class S, typename... A>
    struct dynamic_of<S<A...>, enable_if_t<is_struct_v<S<A...>>>> { using type = S<typename dynamic_of<A>::type...>; };

    template <typename T, size_t N>
    struct dynamic_of<std::array<T, N>, enable_if_t<is_struct_v<std::array<T, N>>>> { using type = std::array<typename dynamic_of<T>::type, N>; };

    template <typename T> constexpr bool is_sliced_v =
        (is_array_v<T> && is_dynamic_v<T>) || (is_struct_v<T> && !std::is_same_v<typename packet_of<T>::type, T>);

    // ---- leaf tables: the device arrays behind a dynamic value, in declaration order --------------------------------
    template <typename D> struct leaf_counter { static constexpr size_t value = dynamic_leaf_count<D>::value; };

    template <size_t K> struct LeafTable {
        void *ptr[K ? K : 1];
        uint32_t stride[K ? K : 1];     // 1: one entry per slice, 0: broadcast of a size-1 array
    };

    /// Host side: walk a dynamic value and note the device pointer of every leaf array.  `Writable`: the kernel stores
    /// through the pointers (results, written-back arguments), so shared buffers are unshared first.
    struct LeafScan {
        size_t slices = 0;      // common slice count of the sliced arguments (0: none seen yet)
        size_t bytes = 0;       // algorithmic bytes: every non-broadcast leaf once per direction
        bool ok = true;
    };

    template <bool Writable, typename D, size_t K> void collect_leaves(D &d, LeafTable<K> &t, size_t &k, LeafScan &scan) {
        using DD = std::remove_const_t<D>;
        if constexpr (is_hip_array<DD>::value) {
            DD &array = const_cast<DD &>(d);
            const size_t n = array.size();
            if (n == 0) { scan.ok = false; return; }
            if (n != 1) {
                if (scan.slices <= 1) scan.slices = n;
                else if (scan.slices != n) scan.ok = false;
            } else if (scan.slices == 0) {
                scan.slices = 1;
            }
            if constexpr (Writable) {
                array.make_unique();
                t.ptr[k] = (void *) array.data();
            } else {
                t.ptr[k] = (void *) const_cast<const DD &>(array).data();
            }
            t.stride[k] = n == 1 ? 0u : 1u;
            if (n != 1) scan.bytes += (Writable ? 2 : 1) * n * sizeof(typename DD::Value);
            ++k;
        } else if constexpr (is_struct_v<DD>) {
            struct_support<DD>::apply(const_cast<DD &>(d), [&](auto &f) { collect_leaves<Writable>(f, t, k, scan); });
        } else if constexpr (is_array_v<DD>) {
            for (size_t c = 0; c < DD::Size; ++c) collect_leaves<Writable>(const_cast<DD &>(d).coeff(c), t, k, scan);
        }
    }

    /// Device side: read slice i of the structure described by `t` as a packet
    template <typename P, size_t K> __device__ inline void load_packet(P &p, const LeafTable<K> &t, size_t &k, size_t i) {
        if constexpr (is_lane<P>::value) {
            using T = scalar_t<P>;
            using Stored = std::conditional_t<std::is_same_v<T, bool>, uint8_t, T>;
            p = P((T) static_cast<const Stored *>(t.ptr[k])[i * t.stride[k]]);
            ++k;
        } else if constexpr (is_struct_v<P>) {
            struct_support<P>::apply(p, [&](auto &f) { load_packet(f, t, k, i); });
        } else {
            for (size_t c = 0; c < P::Size; ++c) load_packet(p.coeff(c), t, k, i);
        }
    }

    template <typename P, size_t K> __device__ inline void store_packet(const P &p, const LeafTable<K> &t, size_t &k, size_t i) {
        if constexpr (std::is_arithmetic_v<P>) {
            using Stored = std::conditional_t<std::is_same_v<P, bool>, uint8_t, P>;
            static_cast<Stored *>(t.ptr[k])[i] = (Stored) p;
            ++k;
        } else if constexpr (is_lane<P>::value) {
            using T = scalar_t<P>;
            using Stored = std::conditional_t<std::is_same_v<T, bool>, uint8_t, T>;
            if (t.stride[k]) static_cast<Stored *>(t.ptr[k])[i] = (Stored) p.coeff(0);
            ++k;
        } else if constexpr (is_struct_v<P>) {
            struct_support<P>::apply(const_cast<P &>(p), [&](auto &f) { store_packet(f, t, k, i); });
        } else {
            for (size_t c = 0; c < P::Size; ++c) store_packet(p.coeff(c), t, k, i);
        }
    }

    /// What the kernel receives for one argument: a leaf table (sliced) or the value itself
    template <typename Arg, bool WriteBack> struct SlicedArg {
        using Packet = typename packet_of<Arg>::type;
        static constexpr bool write_back = WriteBack;
        LeafTable<leaf_counter<Arg>::value> table;
        __device__ Packet load(size_t i) const { Packet p; size_t k = 0; load_packet(p, table, k, i); return p; }
        __device__ void store(size_t i, const Packet &p) const { size_t k = 0; store_packet(p, table, k, i); }
    };
    template <typename Arg> struct ValueArg {
        using Packet = Arg;
        static constexpr bool write_back = false;
        Arg value;
        __device__ Packet load(size_t) const { return value; }
        __device__ void store(size_t, const Packet &) const { }
    };

}

namespace detail {
    /// bytes per slice that the NEXT vectorize() call moves through pointers captured by its functor (see below)
    inline size_t &vectorize_indirect_bytes_slot() { static thread_local size_t bytes = 0; return bytes; }

    template <typename Func, typename OutTable, typename ResultPacket, typename... Views>
    __global__ __launch_bounds__(256) void k_vectorize(size_t n, Func f, OutTable out, Views... views) {
        const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
        if (i >= n) return;
        [&](typename Views::Packet... packets) {
            if constexpr (std::is_void_v<ResultPacket>) {
                f(packets...);
            } else {
                ResultPacket r = f(packets...);
                size_t k = 0;
                store_packet(r, out, k, i);
            }
            ((Views::write_back ? views.store(i, packets) : void()), ...);
        }(views.load(i)...);
    }

    template <typename Arg> struct view_of {
        using A = std::remove_reference_t<Arg>;
        using D = std::remove_const_t<A>;
        static constexpr bool sliced = is_sliced_v<D>;
        static constexpr bool write_back = sliced && std::is_lvalue_reference_v<Arg> && !std::is_const_v<A>;
        using type = std::conditional_t<sliced, SlicedArg<D, write_back>, ValueArg<D>>;
    };

    template <typename Arg> typename view_of<Arg>::type make_view(Arg &&arg, LeafScan &scan) {
        using V = view_of<Arg>;
        if constexpr (V::sliced) {
            typename V::type v;
            size_t k = 0;
            collect_leaves<V::write_back>(arg, v.table, k, scan);
            return v;
        } else {
            return typename V::type{ arg };
        }
    }
}

ENOKI_DEVICE_CODE_END

/// See the top of this file
ENOKI_DEVICE_CODE_BEGIN      // callable from user functions that are themselves bracketed (it only ever RUNS on the host)

template <bool Resize = false, typename Func, typename... Args>
auto vectorize(Func &&f, Args &&... args)
    -> typename detail::dynamic_of<decltype(f(std::declval<typename detail::view_of<Args>::type::Packet &>()...))>::type {
    using ResultPacket = decltype(f(std::declval<typename detail::view_of<Args>::type::Packet &>()...));
    using Result = typename detail::dynamic_of<ResultPacket>::type;
    using F = std::decay_t<Func>;
    using OutTable = detail::LeafTable<detail::leaf_counter<std::conditional_t<std::is_void_v<Result>, int, Result>>::value>;
    auto kernel = &detail::k_vectorize<F, OutTable, ResultPacket, typename detail::view_of<Args>::type...>;
#if defined(__HIP_DEVICE_COMPILE__)
    // device pass of the same translation unit: only the kernel above has to be emitted; this body never runs
    (void) kernel;
    if constexpr (!std::is_void_v<Result>) return Result();
#else
    detail::LeafScan scan;
    scan.bytes = 0;
    auto views = std::make_tuple(detail::make_view<Args>(std::forward<Args>(args), scan)...);
    if (!scan.ok) throw std::runtime_error("vectorize(): vector arguments have incompatible lengths");
    const size_t slice_count = scan.slices ? scan.slices : 1;
    scan.bytes += detail::vectorize_indirect_bytes_slot() * slice_count;     // gathers / scatters inside f (declared by the caller)
    detail::vectorize_indirect_bytes_slot() = 0;

    detail::hip_check(ek_hip_init(-1), "vectorize");
    hipStream_t stream = (hipStream_t) ek_hip_stream();
    const unsigned grid = (unsigned) ((slice_count + 255) / 256);

    OutTable out{};
    if constexpr (std::is_void_v<Result>) {
        std::apply([&](auto &... v) { hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, stream, slice_count, f, out, v...); }, views);
        detail::hip_check(ek_hip_note_launch("vectorize", slice_count, scan.bytes), "vectorize");
    } else {
        Result result = empty<Result>(slice_count);
        detail::LeafScan out_scan;
        size_t k = 0;
        detail::collect_leaves<false>(result, out, k, out_scan);          // freshly allocated: nothing shares these buffers
        for (size_t j = 0; j < k; ++j) out.stride[j] = 1;                 // results are always stored, also when there is ONE slice
        std::apply([&](auto &... v) { hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, stream, slice_count, f, out, v...); }, views);
        detail::hip_check(ek_hip_note_launch("vectorize", slice_count, scan.bytes + out_scan.bytes), "vectorize");
        return result;
    }
#endif
}

/// Accounting only: the next vectorize() call of this thread gathers / scatters `bytes_per_slice` bytes per slice through
/// raw pointers captured by its functor (the kernel cannot know); they are added to the algorithmic bytes that the launch
/// reports to ek_hip_profile_* / bench.py.
inline void vectorize_indirect_bytes(size_t bytes_per_slice) { detail::vectorize_indirect_bytes_slot() = bytes_per_slice; }

template <typename Func, typename... Args> auto vectorize_safe(Func &&f, Args &&... args) {
    return vectorize<true>(std::forward<Func>(f), std::forward<Args>(args)...);
}

ENOKI_DEVICE_CODE_END

} // namespace enoki
