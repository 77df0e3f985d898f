/*
    enoki/fwd.h -- forward declarations of the library's types (reference: include/enoki/fwd.h)

    For headers of a project that name array types in signatures without needing their definitions.  Default template
    arguments live with the definitions (enoki/array.h ...), so a declaration here never conflicts with them.
*/
#pragma once

#include <cstddef>
#include <cstdint>

namespace enoki {

struct ArrayTag;
struct half;

template <typename Value, size_t Size> struct Array;            // static array: Size components side by side
template <typename Value> struct HIPArray;                      // device array (the role of CUDAArray)
template <typename Type> struct DiffArray;                      // differentiable wrapper
template <typename Type> struct Tape;                           // the graph behind DiffArray

template <typename Value, size_t Size> struct Matrix;
template <typename Value> struct Complex;
template <typename Value> struct Quaternion;
template <typename T> struct PCG32;

template <typename T> struct divisor;
template <typename T> struct divisor_ext;
template <typename T, typename> struct struct_support;

// (the reference's names CUDAArray<T> / DynamicArray<Packet<T>> are aliases defined in enoki/cuda.h / enoki/dynamic.h)

} // namespace enoki
