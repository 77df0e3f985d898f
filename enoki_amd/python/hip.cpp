// enoki_amd.hip -- non-differentiable device arrays (the analogue of the reference's enoki.cuda module,
// src/python/cuda.cpp:14-53 + cuda_1d.cpp:4-107)
#include "common.h"
#include <enoki/random.h>

using FloatC = HIPArray<float>;
using DoubleC = HIPArray<double>;
using Int32C = HIPArray<int32_t>;
using UInt32C = HIPArray<uint32_t>;
using Int64C = HIPArray<int64_t>;
using UInt64C = HIPArray<uint64_t>;
using MaskC = HIPArray<bool>;

PYBIND11_MODULE(hip, m) {
    m.doc() = "MI355X-native Enoki arrays (eager HIP kernels, no JIT)";
    bind_runtime(m);
    auto mask = bind_array<MaskC>(m, "Mask");
    auto f32 = bind_array<FloatC>(m, "Float32");
    auto f64 = bind_array<DoubleC>(m, "Float64");
    auto i32 = bind_array<Int32C>(m, "Int32");
    auto u32 = bind_array<UInt32C>(m, "UInt32");
    auto i64 = bind_array<Int64C>(m, "Int64");
    auto u64 = bind_array<UInt64C>(m, "UInt64");
    m.attr("Float") = m.attr("Float32");
    bind_vector_family<HIPArray>(m);             // Vector{0..4}{m,i,u,f,d}
    bind_matrix<FloatC, 2>(m, "Matrix2f"); bind_matrix<DoubleC, 2>(m, "Matrix2d");
    bind_matrix<FloatC, 3>(m, "Matrix3f"); bind_matrix<DoubleC, 3>(m, "Matrix3d");
    bind_matrix<FloatC, 4>(m, "Matrix4f"); bind_matrix<DoubleC, 4>(m, "Matrix4d");
    bind_complex<FloatC>(m, "Complex2f"); bind_complex<DoubleC>(m, "Complex2d");
    bind_quaternion<FloatC>(m, "Quaternion4f"); bind_quaternion<DoubleC>(m, "Quaternion4d");
    m.def("meshgrid", [](const FloatC &x, const FloatC &y) { return meshgrid(x, y); });
    m.def("meshgrid", [](const DoubleC &x, const DoubleC &y) { return meshgrid(x, y); });
    // groups of lanes holding the same 64-bit value (instance pointers), ascending by value: [(value, UInt32 lanes), ...]
    // (cuda_1d.cpp:104-106 over cuda_partition; here the sort-based partition_ of enoki/array_call.h)
    m.def("partition", [](const UInt64C &x) {
        HIPArray<void *> pointers(x);
        std::vector<std::pair<uint64_t, UInt32C>> groups;
        for (const auto &g : pointers.partition_()) groups.emplace_back((uint64_t) (uintptr_t) g.first, g.second);
        return groups;
    });

    bind_cast<FloatC, Int32C>(f32); bind_cast<FloatC, UInt32C>(f32); bind_cast<FloatC, DoubleC>(f32);
    bind_cast<FloatC, Int64C>(f32); bind_cast<FloatC, UInt64C>(f32);
    bind_cast<DoubleC, FloatC>(f64); bind_cast<DoubleC, Int32C>(f64); bind_cast<DoubleC, UInt32C>(f64);
    bind_cast<Int32C, FloatC>(i32); bind_cast<Int32C, UInt32C>(i32); bind_cast<Int32C, Int64C>(i32);
    bind_cast<UInt32C, FloatC>(u32); bind_cast<UInt32C, Int32C>(u32); bind_cast<UInt32C, UInt64C>(u32);
    bind_cast<Int64C, Int32C>(i64); bind_cast<Int64C, FloatC>(i64); bind_cast<Int64C, UInt64C>(i64);
    bind_cast<UInt64C, UInt32C>(u64); bind_cast<UInt64C, FloatC>(u64); bind_cast<UInt64C, Int64C>(u64);

    bind_memory<FloatC, UInt32C>(m); bind_memory<FloatC, Int32C>(m);
    bind_memory<Int32C, UInt32C>(m); bind_memory<UInt32C, UInt32C>(m);
    bind_memory<DoubleC, UInt32C>(m);
    bind_memory<FloatC, UInt64C>(m); bind_memory<FloatC, Int64C>(m);      // 64-bit index arrays (narrowed once, enoki/hip.h)

    // PCG32 (src/python/random.h:9-80, cuda_pcg32.cpp)
    using RNG = PCG32<FloatC>;
    py::class_<RNG>(m, "PCG32")
        .def(py::init<UInt64C, UInt64C>(), "initstate"_a = UInt64C(uint64_t(PCG32_DEFAULT_STATE)),
             "initseq"_a = UInt64C(uint64_t(PCG32_DEFAULT_STREAM)))
        .def("seed", &RNG::seed, "initstate"_a = UInt64C(uint64_t(PCG32_DEFAULT_STATE)),
             "initseq"_a = UInt64C(uint64_t(PCG32_DEFAULT_STREAM)))
        .def("__sub__", [](const RNG &a, const RNG &b) { return a - b; })
        .def("__eq__", [](const RNG &a, const RNG &b) { return a == b; })
        .def("__ne__", [](const RNG &a, const RNG &b) { return a != b; })
        .def("advance", &RNG::advance, "delta"_a)
        .def("next_uint32", [](RNG &r) { return r.next_uint32(); })
        .def("next_uint32", [](RNG &r, const MaskC &mk) { return r.next_uint32(mk); }, "mask"_a)
        .def("next_uint64", [](RNG &r) { return r.next_uint64(); })
        .def("next_uint64", [](RNG &r, const MaskC &mk) { return r.next_uint64(mk); }, "mask"_a)
        .def("next_uint32_bounded", [](RNG &r, uint32_t bound) { return r.next_uint32_bounded(bound); }, "bound"_a)
        .def("next_uint32_bounded", [](RNG &r, uint32_t bound, const MaskC &mk) { return r.next_uint32_bounded(bound, mk); },
             "bound"_a, "mask"_a)
        .def("next_uint64_bounded", [](RNG &r, uint64_t bound) { return r.next_uint64_bounded(bound); }, "bound"_a)
        .def("next_uint64_bounded", [](RNG &r, uint64_t bound, const MaskC &mk) { return r.next_uint64_bounded(bound, mk); },
             "bound"_a, "mask"_a)
        .def("next_float32", [](RNG &r) { return r.next_float32(); })
        .def("next_float32", [](RNG &r, const MaskC &mk) { return r.next_float32(mk); }, "mask"_a)
        .def("next_float64", [](RNG &r) { return r.next_float64(); })
        .def("next_float64", [](RNG &r, const MaskC &mk) { return r.next_float64(mk); }, "mask"_a)
        .def_readwrite("state", &RNG::state)
        .def_readwrite("inc", &RNG::inc);
}
