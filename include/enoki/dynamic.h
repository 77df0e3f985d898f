/*
    enoki/dynamic.h -- source compatibility for programs written against the reference's CPU dynamic arrays

        using FloatP = Packet<float>;  using FloatX = DynamicArray<FloatP>;  using FloatD = DiffArray<FloatX>;

    (the aliases of the reference's tests/autodiff.cpp:19-22) become HIPArray<float> / DiffArray<HIPArray<float>>: `Packet`
    only carries the element type and a nominal width, `DynamicArray<Packet<T>>` is the device array of T -- exactly the
    substitution the reference makes for its own GPU backend.  In a hipcc translation unit the header also brings in
    enoki::vectorize() (include/enoki/vectorize.h), whose kernels run the user's packet code on one-element packets.
    tests/cpp/reftest_autodiff_hip.cpp and reftest_sphere_hip.cpp compile the reference's own test sources through it.
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

template <typename P> using DynamicArray = HIPArray<typename P::Value>;

} // namespace enoki
