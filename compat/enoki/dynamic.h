/*
    enoki/dynamic.h -- source compatibility for programs written against the reference's CPU dynamic arrays

        using FloatP = Packet<float>;  using FloatX = DynamicArray<FloatP>;  using FloatD = DiffArray<FloatX>;

    (the aliases of the reference's tests/autodiff.cpp:19-22) become HIPArray<float> / DiffArray<HIPArray<float>>: `Packet`
    only carries the element type and a nominal width, `DynamicArray<Packet<T>>` is the device array of T -- exactly the
    substitution the reference makes for its own GPU backend.  In a hipcc translation unit the header also brings in
    enoki::vectorize() (include/enoki/vectorize.h), whose kernels run the user's packet code on one-element packets.
    tests/cpp/reftest_autodiff_hip.cpp and reftest_sphere_hip.cpp compile the reference's own test sources through it.

    OPT-IN.  In the reference `DynamicArray<Packet<T>>` is a HOST array (include/enoki/dynamic.h:54-964: SoA packets in
    main memory).  A translation unit that includes this header for host data must not silently receive device arrays, so the
    substitution only happens when the build says so: compile with -DENOKI_HIP_DYNAMIC_IS_DEVICE (the retargeted reference
    tests do).  Without it the name exists, and any use of it stops the compilation with an explanation.  There is no CPU
    array type in this library and no fallback.
*/
#pragma once

#if defined(__HIP__)
#  include <enoki/vectorize.h>      // first: makes the array vocabulary callable from vectorize() kernels
#endif
#include <enoki/hip.h>
#include <enoki/array_call.h>

namespace enoki {

template <typename T, size_t N = 16> struct Packet {
    using Value = T;
    static constexpr size_t Size = N;
};

#if defined(ENOKI_HIP_DYNAMIC_IS_DEVICE)
template <typename P> using DynamicArray = HIPArray<typename P::Value>;
#else
namespace detail { template <typename> struct hip_dynamic_opt_in : std::false_type { }; }
template <typename P> struct DynamicArray {
    static_assert(detail::hip_dynamic_opt_in<P>::value,
                  "enoki/dynamic.h of the MI355X backend: DynamicArray<Packet<T>> is the reference's HOST array "
                  "(reference include/enoki/dynamic.h:54-60); this library has device arrays only.  Define "
                  "ENOKI_HIP_DYNAMIC_IS_DEVICE to make DynamicArray<Packet<T>> an alias of HIPArray<T> (device memory), "
                  "or use enoki::HIPArray<T> directly.");
};
#endif

} // namespace enoki
