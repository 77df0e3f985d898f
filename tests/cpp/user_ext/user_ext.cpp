// A DOWNSTREAM extension module, the way a renderer built on the library binds its own functions: it includes the array
// headers and pybind11, does NOT link or include anything of enoki_amd/python/, and takes / returns the array types that
// `enoki_amd.hip` and `enoki_amd.hip_autodiff` registered (pybind11's registry is process wide; the reference's users do the
// same with `enoki.cuda` types -- src/python/common.h registers them globally for that purpose).
//
//     g++ -O2 -std=c++17 -fPIC -shared -fvisibility=hidden -I<pybind11> -I<python> -Iinclude tests/cpp/user_ext/user_ext.cpp \
//         -o tests/cpp/user_ext/user_ext<EXT_SUFFIX> -Lenoki_amd -lenoki-hip-autodiff -lenoki-hip
#include <enoki/hip.h>
#include <enoki/autodiff.h>
#include <enoki/python.h>          // NumPy <-> host static arrays

#include <pybind11/pybind11.h>

namespace py = pybind11;
using namespace enoki;

using FloatC = HIPArray<float>;
using UInt32C = HIPArray<uint32_t>;
using MaskC = HIPArray<bool>;
using Vector3fC = Array<FloatC, 3>;
using FloatD = DiffArray<FloatC>;
using Vector3fD = Array<FloatD, 3>;

/// a templated kernel of the downstream project, instantiated on the device and on the differentiable type
template <typename Vector3> auto shade(const Vector3 &n, const Vector3 &light, const value_t<Vector3> &albedo) {
    return albedo * max(dot(normalize(n), light), 0.f);
}

PYBIND11_MODULE(user_ext, m) {
    m.def("saxpy", [](float a, const FloatC &x, const FloatC &y) { return fmadd(FloatC(a), x, y); });
    m.def("shade", [](const Vector3fC &n, const Vector3fC &l, const FloatC &albedo) { return shade(n, l, albedo); });
    m.def("shade", [](const Vector3fD &n, const Vector3fD &l, const FloatD &albedo) { return shade(n, l, albedo); });
    m.def("lookup", [](const Vector3fC &table, const UInt32C &index, const MaskC &mask) { return gather<Vector3fC>(table, index, mask); });
    m.def("count_positive", [](const FloatC &x) { return count(x > 0.f); });
    // host static arrays travel as NumPy arrays (enoki/python.h), innermost dimension first
    using Vector3f = Array<float, 3>;
    m.def("reflect", [](const Vector3f &d, const Vector3f &n) { return d - n * (2.f * dot(d, n)); });
    m.def("outer", [](const Vector3f &a, const Array<float, 2> &b) {
        Array<Vector3f, 2> r(a * b.x(), a * b.y());              // shape (3, 2) on the NumPy side
        return r;
    });
    m.def("trace", [](const Array<Array<double, 2>, 2> &m2) { return m2.coeff(0).coeff(0) + m2.coeff(1).coeff(1); });
    m.def("positive", [](const Vector3f &v) { return v > 0.f; });
}
