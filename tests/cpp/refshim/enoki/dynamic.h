// tests/cpp/refshim/enoki/dynamic.h -- what the reference's tests get when they `#include <enoki/dynamic.h>` in the
// retargeted build: the CPU array template names, mapped onto the device types of this repository.
//
//     using FloatP = Packet<float>;  using FloatX = DynamicArray<FloatP>;  using FloatD = DiffArray<FloatX>;
//
// (tests/autodiff.cpp:19-22) thereby become HIPArray<float> and DiffArray<HIPArray<float>>, exactly the substitution the
// reference itself makes for its CUDA backend.  `Packet` only carries the element type and a nominal width; tests that
// spell their packets `Array<float>` (tests/sphere.cpp) get this repository's one-element packet, the unit that
// vectorize() instantiates kernels on.
#pragma once

#if defined(__HIP__)
#  include <enoki/vectorize.h>      // must come first: makes the array vocabulary callable from vectorize() kernels
#endif
#include <enoki/hip.h>
#include <enoki/array_call.h>

#include <iostream>

namespace enoki {

template <typename T, size_t N = 16> struct Packet {
    using Value = T;
    static constexpr size_t Size = N;
};

template <typename P> using DynamicArray = HIPArray<typename P::Value>;

} // namespace enoki

#include <test_assert.h>
