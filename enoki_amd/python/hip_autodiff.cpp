// enoki_amd.hip_autodiff -- differentiable device arrays (the analogue of enoki.cuda_autodiff,
// src/python/cuda_autodiff.cpp:13-31 + cuda_autodiff_1d.cpp:4-156)
#include "common.h"

using FloatC = HIPArray<float>;
using FloatD = DiffArray<HIPArray<float>>;
using DoubleC = HIPArray<double>;
using DoubleD = DiffArray<HIPArray<double>>;
using Int32D = DiffArray<HIPArray<int32_t>>;
using UInt32D = DiffArray<HIPArray<uint32_t>>;
using Int64D = DiffArray<HIPArray<int64_t>>;
using UInt64D = DiffArray<HIPArray<uint64_t>>;
using MaskD = DiffArray<HIPArray<bool>>;
template <typename T> using DiffHIP = DiffArray<HIPArray<T>>;

PYBIND11_MODULE(hip_autodiff, m) {
    m.doc() = "MI355X-native differentiable Enoki arrays (tape-based reverse/forward mode)";
    py::module_::import("enoki_amd.hip");      // plain array classes (gradient(), detach() return them)
    bind_runtime(m);
    auto mask = bind_array<MaskD>(m, "Mask");
    auto f32 = bind_array<FloatD>(m, "Float32");
    auto f64 = bind_array<DoubleD>(m, "Float64");       // Tape<HIPArray<double>> (autodiff.cpp:1240 analogue)
    auto i32 = bind_array<Int32D>(m, "Int32");
    auto u32 = bind_array<UInt32D>(m, "UInt32");
    auto i64 = bind_array<Int64D>(m, "Int64");          // cuda_autodiff_1d.cpp:56-97 binds the 64-bit integers too
    auto u64 = bind_array<UInt64D>(m, "UInt64");
    m.attr("Float") = m.attr("Float32");
    bind_vector_family<DiffHIP>(m);              // Vector{0..4}{m,i,u,f,d} (cuda_autodiff_{0..4}d.cpp)
    bind_matrix<FloatD, 2>(m, "Matrix2f"); bind_matrix<DoubleD, 2>(m, "Matrix2d");
    bind_matrix<FloatD, 3>(m, "Matrix3f"); bind_matrix<DoubleD, 3>(m, "Matrix3d");
    bind_matrix<FloatD, 4>(m, "Matrix4f"); bind_matrix<DoubleD, 4>(m, "Matrix4d");
    bind_complex<FloatD>(m, "Complex2f"); bind_complex<DoubleD>(m, "Complex2d");
    bind_quaternion<FloatD>(m, "Quaternion4f"); bind_quaternion<DoubleD>(m, "Quaternion4d");

    f32.def(py::init([](const FloatC &v) { return FloatD(v); }));
    f64.def(py::init([](const DoubleC &v) { return DoubleD(v); }));
    bind_cast<FloatD, DoubleD>(f32); bind_cast<DoubleD, FloatD>(f64);
    u32.def(py::init([](const HIPArray<uint32_t> &v) { return UInt32D(v); }));
    i32.def(py::init([](const HIPArray<int32_t> &v) { return Int32D(v); }));
    mask.def(py::init([](const HIPArray<bool> &v) { return MaskD(v); }));
    bind_cast<FloatD, Int32D>(f32); bind_cast<FloatD, UInt32D>(f32);
    bind_cast<Int32D, FloatD>(i32); bind_cast<Int32D, UInt32D>(i32);
    bind_cast<UInt32D, FloatD>(u32); bind_cast<UInt32D, Int32D>(u32);

    i64.def(py::init([](const HIPArray<int64_t> &v) { return Int64D(v); }));
    u64.def(py::init([](const HIPArray<uint64_t> &v) { return UInt64D(v); }));
    bind_cast<FloatD, Int64D>(f32); bind_cast<FloatD, UInt64D>(f32);
    bind_cast<Int64D, Int32D>(i64); bind_cast<Int64D, FloatD>(i64); bind_cast<Int64D, UInt64D>(i64);
    bind_cast<UInt64D, UInt32D>(u64); bind_cast<UInt64D, FloatD>(u64); bind_cast<UInt64D, Int64D>(u64);
    bind_cast<Int32D, Int64D>(i32); bind_cast<UInt32D, UInt64D>(u32);
    m.def("meshgrid", [](const FloatD &x, const FloatD &y) { return meshgrid(x, y); });
    m.def("meshgrid", [](const DoubleD &x, const DoubleD &y) { return meshgrid(x, y); });

    bind_memory<FloatD, UInt32D>(m); bind_memory<FloatD, Int32D>(m);
    bind_memory<UInt32D, UInt32D>(m); bind_memory<Int32D, UInt32D>(m);
    bind_memory<DoubleD, UInt32D>(m);
    bind_memory<FloatD, UInt64D>(m); bind_memory<FloatD, Int64D>(m);      // 64-bit index arrays (narrowed once, enoki/hip.h)
}
