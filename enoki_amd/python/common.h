/*
    python/common.h -- generic pybind11 binder for 1-D device arrays

    Mirrors the surface of the reference's `bind<Array>()` (src/python/common.h:338-998) for the types on
    the hot path: each array class gets constructors (scalar, numpy, copy), arithmetic / comparison
    operators, zero/empty/full/arange/linspace, numpy()/torch interop, and the module gets the free
    functions (fmadd, sin, ..., gather, scatter, scatter_add, hsum, ..., select, backward, gradient, ...)
    overloaded per array type -- so `import enoki_amd.hip_autodiff as ek; ek.hsum(ek.sin(ek.fmadd(a, x, b)))`
    reads like the reference's `enoki.cuda_autodiff`.
*/
#pragma once

#include <pybind11/functional.h>
#include <pybind11/numpy.h>
#include <pybind11/operators.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <unordered_map>

#include <enoki/hip.h>
#include <enoki/autodiff.h>
#include <enoki/array_call.h>
#include <enoki/matrix.h>
#include <enoki/morton.h>
#include <enoki/sh.h>
#include <enoki/transform.h>
#include <enoki/special.h>
#include <enoki/complex.h>
#include <enoki/quaternion.h>

#include <sstream>

namespace py = pybind11;
using namespace py::literals;
using namespace enoki;

template <typename Array> std::string array_repr(const Array &a) {
    const auto &v = detach(a);
    auto host = v.to_host();
    std::ostringstream oss;
    oss << "[";
    size_t n = host.size(), shown = 0;
    for (size_t i = 0; i < n; ++i) {
        if (n > 20 && i == 5) { oss << ".. " << (n - 10) << " skipped .., "; i = n - 6; continue; }
        oss << (is_mask_v<Array> ? (double) (host[i] != 0) : (double) host[i]);
        if (i + 1 < n) oss << ", ";
        ++shown;
    }
    oss << "]";
    return oss.str();
}

/// Make work queued on torch's current stream and on the library stream visible to each other.  When the
/// library has adopted torch's stream (enoki_amd.dist.adopt_torch_stream) nothing needs to happen.
inline void sync_with_torch() {
    py::object sys_modules = py::module_::import("sys").attr("modules");
    if (!sys_modules.contains("torch"))
        return;
    py::object cuda = py::module_::import("torch").attr("cuda");
    if (!cuda.attr("is_initialized")().cast<bool>())
        return;
    uintptr_t torch_stream = cuda.attr("current_stream")().attr("cuda_stream").cast<uintptr_t>();
    if (torch_stream == (uintptr_t) ek_hip_stream())
        return;
    cuda.attr("current_stream")().attr("synchronize")();
    detail::hip_check(ek_hip_sync(), "hip_sync");
}

/// Bind one 1-D array type.  `Mask` / `UInt32` / `Int32` are the sibling types of the same module.
template <typename Array> py::class_<Array> bind_array(py::module_ &m, const char *name) {
    using Scalar = scalar_t<Array>;
    using Plain = std::decay_t<decltype(detach(std::declval<const Array &>()))>;     // HIPArray<Scalar>
    using Store = std::conditional_t<std::is_same_v<Scalar, bool>, uint8_t, Scalar>;
    using Mask = mask_t<Array>;
    constexpr bool IsMask = is_mask_v<Array>;
    constexpr bool IsFloat = std::is_floating_point_v<Scalar>;
    constexpr bool IsInt = std::is_integral_v<Scalar> && !IsMask;
    constexpr bool IsDiff = is_diff_array_v<Array>;

    py::class_<Array> cl(m, name);
    cl.def(py::init<>())
      .def(py::init<const Array &>())
      .def(py::init<Scalar>())
      .def(py::init([](py::array_t<Store, py::array::c_style | py::array::forcecast> a) {
          return Array(Plain::copy(a.data(), (size_t) a.size()));
      }))
      .def(py::init([](py::object o) {
          // any object with __cuda_array_interface__ (a torch ROCm tensor, cupy, another module's array): one
          // device-to-device copy, no host round trip (the reference goes through a scatter, common.h:1067-1161)
          if (py::isinstance<Plain>(o))
              throw py::reference_cast_error();     // our own plain array: the sharing constructor handles it (no copy)
          if (py::hasattr(o, "requires_grad") && py::hasattr(o, "detach") && o.attr("requires_grad").cast<bool>())
              o = o.attr("detach")();          // torch refuses to export tensors that require grad
          if (py::hasattr(o, "is_contiguous") && !o.attr("is_contiguous")().cast<bool>())
              o = o.attr("contiguous")();
          // "not mine": let pybind11 try the remaining overloads (the converting constructors registered later)
          if (!py::hasattr(o, "__cuda_array_interface__"))
              throw py::reference_cast_error();
          py::dict d = o.attr("__cuda_array_interface__");
          std::string typestr = d["typestr"].cast<std::string>();
          char kind = IsFloat ? 'f' : (IsMask ? 'b' : (std::is_unsigned_v<Scalar> ? 'u' : 'i'));
          bool ok = typestr.size() == 3 && typestr[1] == kind && typestr[2] == char('0' + sizeof(Store));
          if (IsMask && typestr.size() == 3 && typestr[1] == 'u' && typestr[2] == '1') ok = true;
          if (!ok) throw py::reference_cast_error();      // other element type: a converting constructor may take it
          py::tuple shape = d["shape"];
          if (shape.size() > 1) throw py::type_error("__cuda_array_interface__: expected a 0-d or 1-d array");
          size_t n = shape.size() == 0 ? 1 : shape[0].cast<size_t>();
          if (d.contains("strides") && !d["strides"].is_none() && shape.size() == 1 && n > 1 &&
              py::tuple(d["strides"])[0].cast<size_t>() != sizeof(Store))
              throw py::type_error("__cuda_array_interface__: array is not contiguous");
          uintptr_t ptr = py::tuple(d["data"])[0].cast<uintptr_t>();
          sync_with_torch();
          Plain r = empty<Plain>(n);
          if (n) detail::hip_check(ek_hip_memcpy_device(r.data(), (const void *) ptr, n * sizeof(Store)), "copy from device object");
          return Array(r);
      }))
      .def("torch", [](Array &a) {
          // zero-copy torch view (the tensor keeps this object alive); the reference's .torch() copies
          sync_with_torch();
          py::object torch = py::module_::import("torch");
          return torch.attr("as_tensor")(py::cast(a, py::return_value_policy::reference), "device"_a = "cuda");
      })
      .def("__len__", [](const Array &a) { return a.size(); })
      .def("__repr__", [](const Array &a) { return array_repr(a); })
      .def("__getitem__", [](const Array &a, size_t i) {
          if (i >= a.size()) throw py::index_error();
          return a.coeff(i);
      })
      .def("__setitem__", [](Array &a, const Mask &m, const Array &v) { masked(a, m) = v; },
           "a[mask] = value: masked assignment (a select, array_base.h:144-157)")
      .def("size", [](const Array &a) { return a.size(); })
      .def("eval", [](Array &a) -> Array & { return a.eval(); }, py::return_value_policy::reference)
      .def("managed", [](Array &a) -> Array & { return a.managed(); }, py::return_value_policy::reference)
      .def("numpy", [](const Array &a) {
          auto host = detach(a).to_host();
          py::array_t<Store> out((py::ssize_t) host.size());
          if (!host.empty()) memcpy(out.mutable_data(), host.data(), host.size() * sizeof(Store));
          return out;
      })
      .def("__array__", [](const Array &a, py::args, py::kwargs) {       // np.asarray(x): a host copy, like .numpy()
          auto host = detach(a).to_host();
          py::array_t<Store> out((py::ssize_t) host.size());
          if (!host.empty()) memcpy(out.mutable_data(), host.data(), host.size() * sizeof(Store));
          return out;
      })
      .def("data_ptr", [](Array &a) { return (uintptr_t) a.data(); }, "raw device pointer")
      .def("explain", [](const Array &a) { return detach(a).explain_(); },
           "what state the array is in (evaluated / which kind of unevaluated node) and what its consumers can still fuse -- "
           "does not evaluate anything; hip_set_log_level(2) additionally prints a line whenever an expression leaves bucket order")
      .def_property_readonly("__cuda_array_interface__", [](Array &a) {
          // consumed by torch.as_tensor(obj, device='cuda') on ROCm builds: zero-copy view.  The consumer keeps the python
          // OBJECT alive, not the buffer: the buffer is marked as exported so that a later copy-on-write (a scatter into
          // this array while another handle shares it) parks the old buffer instead of letting it return to the allocator
          // under the view -- such a view goes stale, it never dangles.
          py::dict d;
          a.data();
          detach(a).mark_exported_();
          char typestr[4] = { '<', IsFloat ? 'f' : (IsMask ? 'b' : (std::is_unsigned_v<Scalar> ? 'u' : 'i')),
                              char('0' + sizeof(Store)), 0 };
          d["shape"] = py::make_tuple(a.size());
          d["typestr"] = std::string(typestr);
          d["data"] = py::make_tuple((uintptr_t) a.data(), false);
          d["version"] = 2;
          return d;
      })
      .def_static("map", [](uintptr_t ptr, size_t size) { return Array(Plain::map((void *) ptr, size, false)); },
                  "ptr"_a, "size"_a, "wrap device memory owned by the caller (e.g. tensor.data_ptr())")
      .def_static("zero", [](size_t size) { return zero<Array>(size); }, "size"_a = 1)
      .def_static("empty", [](size_t size) { return empty<Array>(size); }, "size"_a = 1)
      .def_static("full", [](Scalar value, size_t size) { return full<Array>(value, size); }, "value"_a, "size"_a = 1);

    // the rest of the reference's per-array surface (src/python/common.h:668-740): iteration over a host copy, a[mask],
    // resize, the raw pointer as a property, shape()
    cl.def("__iter__", [](const Array &a) {
          auto host = detach(a).to_host();
          py::list values;
          for (auto v : host) values.append(py::cast((Scalar) v));
          return py::iter(values);
      })
      .def("__getitem__", [](const Array &a, const Mask &mk) { return select(mk, a, Array(Scalar(0))); },
           "a[mask]: the active entries, zero elsewhere (a select, common.h:699-701)")
      .def("resize", [](Array &a, size_t size) { a.resize(size); })
      .def_property_readonly("data", [](Array &a) { return (uintptr_t) a.data(); });
    if constexpr (IsDiff)
        cl.def_property_readonly("index", [](const Array &a) { return a.index_(); }, "index of the array's node on the tape (0: none)");
    m.def("shape", [](const Array &a) { return std::vector<size_t>{ a.size() }; });

    m.def("slices", [](const Array &a) { return slices(a); });
    m.def("set_slices", [](Array &a, size_t n) { set_slices(a, n); });
    m.def("detach", [](const Array &a) { return Plain(detach(a)); });

    cl.def("__eq__", [](const Array &a, const Array &b) { return eq(a, b); })
      .def("__ne__", [](const Array &a, const Array &b) { return neq(a, b); });
    m.def("eq", [](const Array &a, const Array &b) { return eq(a, b); });
    m.def("neq", [](const Array &a, const Array &b) { return neq(a, b); });
    m.def("select", [](const Mask &mk, const Array &t, const Array &f) { return select(mk, t, f); });

    if constexpr (IsMask) {
        cl.def("__and__", [](const Array &a, const Array &b) { return Array(a & b); })
          .def("__or__", [](const Array &a, const Array &b) { return Array(a | b); })
          .def("__xor__", [](const Array &a, const Array &b) { return Array(a ^ b); })
          .def("__invert__", [](const Array &a) { return Array(!a); });
        m.def("all", [](const Array &a) { return all(a); });
        m.def("any", [](const Array &a) { return any(a); });
        m.def("none", [](const Array &a) { return none(a); });
        m.def("count", [](const Array &a) { return count(a); });
        // flat arrays have one nesting level: the *_nested forms coincide with the plain ones (array_router.h:1331-1383)
        m.def("all_nested", [](const Array &a) { return all(a); });
        m.def("any_nested", [](const Array &a) { return any(a); });
        m.def("none_nested", [](const Array &a) { return none(a); });
        m.def("count_nested", [](const Array &a) { return count(a); });
    } else {
        cl.def_static("arange", [](size_t size) { return arange<Array>(size); }, "size"_a)
          .def(py::self + py::self).def(py::self - py::self).def(py::self * py::self)
          .def(Scalar() + py::self).def(Scalar() - py::self).def(Scalar() * py::self)
          .def(py::self + Scalar()).def(py::self - Scalar()).def(py::self * Scalar())
          .def(-py::self)
          .def("__lt__", [](const Array &a, const Array &b) { return a < b; })
          .def("__le__", [](const Array &a, const Array &b) { return a <= b; })
          .def("__gt__", [](const Array &a, const Array &b) { return a > b; })
          .def("__ge__", [](const Array &a, const Array &b) { return a >= b; })
          .def("__and__", [](const Array &a, const Mask &mk) { return Array(a & mk); });
        m.def("abs", [](const Array &a) { return abs(a); });
        m.def("min", [](const Array &a, const Array &b) { return min(a, b); });
        m.def("max", [](const Array &a, const Array &b) { return max(a, b); });
        m.def("sqr", [](const Array &a) { return sqr(a); });
        m.def("fmadd", [](const Array &a, const Array &b, const Array &c) { return fmadd(a, b, c); });
        m.def("fmsub", [](const Array &a, const Array &b, const Array &c) { return fmsub(a, b, c); });
        m.def("fnmadd", [](const Array &a, const Array &b, const Array &c) { return fnmadd(a, b, c); });
        m.def("fnmsub", [](const Array &a, const Array &b, const Array &c) { return fnmsub(a, b, c); });
        m.def("hsum", [](const Array &a) { return hsum(a); });
        m.def("hprod", [](const Array &a) { return hprod(a); });
        m.def("hmin", [](const Array &a) { return hmin(a); });
        m.def("hmax", [](const Array &a) { return hmax(a); });
        m.def("hsum_nested", [](const Array &a) { return hsum(a); });
        m.def("hprod_nested", [](const Array &a) { return hprod(a); });
        m.def("hmin_nested", [](const Array &a) { return hmin(a); });
        m.def("hmax_nested", [](const Array &a) { return hmax(a); });
        m.def("psum", [](const Array &a) { return psum(a); });
        if constexpr (!IsDiff)
            m.def("compress", [](const Array &a, const Mask &mk) { return compress(a, mk); });
        m.def("reverse", [](const Array &a) { return reverse(a); });
    }

    if constexpr (IsFloat) {
        cl.def_static("linspace", [](Scalar lo, Scalar hi, size_t size) { return linspace<Array>(lo, hi, size); },
                      "min"_a, "max"_a, "size"_a)
          .def(py::self / py::self).def(Scalar() / py::self).def(py::self / Scalar());
        cl.def("__truediv__", [](const Array &a, const Array &b) { return a / b; });
        m.def("sqrt", [](const Array &a) { return sqrt(a); });
        m.def("rcp", [](const Array &a) { return rcp(a); });
        m.def("rsqrt", [](const Array &a) { return rsqrt(a); });
        m.def("floor", [](const Array &a) { return floor(a); });
        m.def("ceil", [](const Array &a) { return ceil(a); });
        m.def("round", [](const Array &a) { return round(a); });
        m.def("trunc", [](const Array &a) { return trunc(a); });
        m.def("sign", [](const Array &a) { return sign(a); });
        m.def("isnan", [](const Array &a) { return isnan(a); });
        m.def("isinf", [](const Array &a) { return isinf(a); });
        m.def("isfinite", [](const Array &a) { return isfinite(a); });
        m.def("safe_sqrt", [](const Array &a) { return safe_sqrt(a); });
        m.def("safe_rsqrt", [](const Array &a) { return safe_rsqrt(a); });
        m.def("safe_asin", [](const Array &a) { return safe_asin(a); });
        m.def("safe_acos", [](const Array &a) { return safe_acos(a); });
        m.def("hypot", [](const Array &a, const Array &b) { return hypot(a, b); });
        m.def("copysign", [](const Array &a, const Array &b) { return copysign(a, b); });
        m.def("copysign_neg", [](const Array &a, const Array &b) { return copysign(a, -b); });
        m.def("mulsign", [](const Array &a, const Array &b) { return mulsign(a, b); });
        m.def("mulsign_neg", [](const Array &a, const Array &b) { return mulsign(a, -b); });
        m.def("hmean", [](const Array &a) { return hsum(a) * Array(Scalar(1) / Scalar(std::max<size_t>(slices(a), 1))); });
        m.def("hmean_nested", [](const Array &a) { return hsum(a) * Array(Scalar(1) / Scalar(std::max<size_t>(slices(a), 1))); });
        // |a - b| <= atol + rtol |b| everywhere (src/python/common.h allclose); NaNs compare unequal unless equal_nan
        m.def("allclose", [](const Array &a, const Array &b, Scalar rtol, Scalar atol, bool equal_nan) {
            auto ok = abs(detach(a) - detach(b)) <= fmadd(abs(detach(b)), std::decay_t<decltype(detach(b))>(rtol), std::decay_t<decltype(detach(b))>(atol));
            if (equal_nan) ok = ok | (isnan(detach(a)) & isnan(detach(b)));
            return all(ok);
        }, "a"_a, "b"_a, "rtol"_a = Scalar(1e-5), "atol"_a = Scalar(1e-8), "equal_nan"_a = false);
        m.def("sin", [](const Array &a) { return sin(a); });
        m.def("cos", [](const Array &a) { return cos(a); });
        m.def("sincos", [](const Array &a) { return sincos(a); });
        m.def("exp", [](const Array &a) { return exp(a); });
        m.def("log", [](const Array &a) { return log(a); });
        {
            m.def("tan", [](const Array &a) { return tan(a); });
            m.def("cot", [](const Array &a) { return cot(a); });
            m.def("csc", [](const Array &a) { return csc(a); });
            m.def("sec", [](const Array &a) { return sec(a); });
            m.def("asin", [](const Array &a) { return asin(a); });
            m.def("acos", [](const Array &a) { return acos(a); });
            m.def("atan", [](const Array &a) { return atan(a); });
            m.def("atan2", [](const Array &y, const Array &x) { return atan2(y, x); }, "y"_a, "x"_a);
            m.def("sinh", [](const Array &a) { return sinh(a); });
            m.def("cosh", [](const Array &a) { return cosh(a); });
            m.def("sincosh", [](const Array &a) { return sincosh(a); });
            m.def("tanh", [](const Array &a) { return tanh(a); });
            m.def("csch", [](const Array &a) { return csch(a); });
            m.def("sech", [](const Array &a) { return sech(a); });
            m.def("coth", [](const Array &a) { return coth(a); });
            m.def("asinh", [](const Array &a) { return asinh(a); });
            m.def("acosh", [](const Array &a) { return acosh(a); });
            m.def("atanh", [](const Array &a) { return atanh(a); });
            m.def("cbrt", [](const Array &a) { return cbrt(a); });
            m.def("erf", [](const Array &a) { return erf(a); });
            m.def("erfc", [](const Array &a) { return erfc(a); });
            m.def("erfinv", [](const Array &a) { return erfinv(a); });
            m.def("i0e", [](const Array &a) { return i0e(a); });
            m.def("dawson", [](const Array &a) { return dawson(a); });
            m.def("erfi", [](const Array &a) { return erfi(a); });
            m.def("lgamma", [](const Array &a) { return lgamma(a); });
            m.def("tgamma", [](const Array &a) { return tgamma(a); });
            // elliptic integrals (special.h:314-672; include/enoki/ellint.h)
            m.def("comp_ellint_1", [](const Array &k) { return comp_ellint_1(k); });
            m.def("comp_ellint_2", [](const Array &k) { return comp_ellint_2(k); });
            m.def("comp_ellint_3", [](const Array &k, const Array &nu) { return comp_ellint_3(k, nu); });
            m.def("ellint_1", [](const Array &phi, const Array &k) { return ellint_1(phi, k); });
            m.def("ellint_2", [](const Array &phi, const Array &k) { return ellint_2(phi, k); });
            m.def("ellint_3", [](const Array &phi, const Array &k, const Array &nu) { return ellint_3(phi, k, nu); });
            m.def("carlson_rf", [](const Array &x, const Array &y, const Array &z) { return carlson_rf(x, y, z); });
            m.def("carlson_rd", [](const Array &x, const Array &y, const Array &z) { return carlson_rd(x, y, z); });
            m.def("carlson_rc", [](const Array &x, const Array &y) { return carlson_rc(x, y); });
            m.def("carlson_rj", [](const Array &x, const Array &y, const Array &z, const Array &r) { return carlson_rj(x, y, z, r); });
            m.def("pow", [](const Array &a, const Array &b) { return pow(a, b); });
            m.def("pow", [](const Array &a, int b) { return pow(a, b); });
            cl.def("__pow__", [](const Array &a, const Array &b) { return pow(a, b); })
              .def("__pow__", [](const Array &a, int b) { return pow(a, b); })
              .def("__pow__", [](const Array &a, Scalar b) { return pow(a, Array(b)); });
            m.def("fmod", [](const Array &a, const Array &b) { return fmod(a, b); });
            m.def("lerp", [](const Array &a, const Array &b, const Array &t) { return lerp(a, b, t); });
            m.def("clamp", [](const Array &v, const Array &lo, const Array &hi) { return clamp(v, lo, hi); });
        }
    }

    if constexpr (IsInt) {
        cl.def("__floordiv__", [](const Array &a, const Array &b) { return a / b; })
          .def("__mod__", [](const Array &a, const Array &b) { return a % b; })
          .def("__and__", [](const Array &a, const Array &b) { return Array(a & b); })
          .def("__or__", [](const Array &a, const Array &b) { return Array(a | b); })
          .def("__xor__", [](const Array &a, const Array &b) { return Array(a ^ b); })
          .def("__invert__", [](const Array &a) { return Array(~a); })
          .def("__lshift__", [](const Array &a, const Array &b) { return a << b; })
          .def("__rshift__", [](const Array &a, const Array &b) { return a >> b; });
        m.def("rol", [](const Array &a, const Array &k) { return rol(a, k); });
        m.def("ror", [](const Array &a, const Array &k) { return ror(a, k); });
        m.def("popcnt", [](const Array &a) { return popcnt(a); });
        m.def("lzcnt", [](const Array &a) { return lzcnt(a); });
        m.def("tzcnt", [](const Array &a) { return tzcnt(a); });
        m.def("mulhi", [](const Array &a, const Array &b) { return mulhi(a, b); });
        // floor(log2(a)) = (bits - 1) - lzcnt(a)  (array_router.h log2i, bound in src/python/common.h:832)
        m.def("log2i", [](const Array &a) { return log2i(a); });
        if constexpr (std::is_same_v<Scalar, uint32_t>)
            // every lane searches [start, end) for the first index where pred(index) is False (cuda_1d.cpp:85-92,
            // cuda_autodiff_1d.cpp; array_utils.h:130-171): pred maps a UInt32 array to a Mask
            m.def("binary_search", [](uint32_t start, uint32_t end, const std::function<Mask(const Array &)> &pred) {
                return binary_search<Array>(start, end, pred);
            }, "start"_a, "end"_a, "pred"_a);
    }

    if constexpr (IsDiff && IsFloat) {
        m.def("requires_gradient", [](const Array &a) { return requires_gradient(a); });
        m.def("set_requires_gradient", [](Array &a, bool value) { set_requires_gradient(a, value); }, "array"_a,
              "value"_a = true);
        m.def("gradient", [](const Array &a) { return Plain(gradient(a)); });
        m.def("gradient_index", [](const Array &a) { return gradient_index(a); });
        m.def("set_gradient", [](Array &a, const Plain &g, bool backward) { a.set_gradient_(g, backward); }, "array"_a,
              "gradient"_a, "backward"_a = true);
        m.def("backward", [](const Array &a, bool free_graph) { backward(a, free_graph); }, "array"_a,
              "free_graph"_a = true);
        m.def("forward", [](const Array &a, bool free_graph) { forward(a, free_graph); }, "array"_a,
              "free_graph"_a = true);
        m.def("reattach", [](Array &a, const Array &b) { reattach(a, b); });
        m.def("graphviz", [](const Array &a) { return graphviz(a); });
        m.def("set_label", [](const Array &a, const char *label) { set_label(a, label); });
        cl.def_static("backward", [](bool free_graph) { Array::backward_static_(free_graph); }, "free_graph"_a = true)
          .def_static("forward", [](bool free_graph) { Array::forward_static_(free_graph); }, "free_graph"_a = true)
          .def_static("whos", []() { return Array::whos_(); })
          .def_static("simplify_graph", []() { Array::simplify_graph_(); })
          .def_static("set_log_level", [](uint32_t level) { Array::set_log_level_(level); })
          .def_static("push_prefix", [](const char *label) { Array::push_prefix_(label); })
          .def_static("pop_prefix", []() { Array::pop_prefix_(); })
          .def_static("log_level", []() { return Array::log_level_(); })
          .def_static("set_graph_simplification", [](bool value) { Array::set_graph_simplification_(value); });
        // `with Float32.Scope("name"): ...` prefixes the labels of the nodes recorded inside (cuda_autodiff_1d.cpp:144-155)
        struct Scope {
            std::string name;
            void enter() { Array::push_prefix_(name.c_str()); }
            void exit(py::handle, py::handle, py::handle) { Array::pop_prefix_(); }
        };
        py::class_<Scope>(cl, "Scope")
            .def(py::init([](const std::string &name) { return Scope{ name }; }))
            .def("__enter__", &Scope::enter)
            .def("__exit__", &Scope::exit);
    }
    return cl;
}

/// gather / scatter / scatter_add for a (value type, index type) pair
template <typename Array, typename Index> void bind_memory(py::module_ &m) {
    using Mask = mask_t<Array>;
    m.def("gather", [](const Array &source, const Index &index, const Mask &mask) {
        return gather<Array>(source, index, mask);
    }, "source"_a, "index"_a, "mask"_a = Mask(true));
    m.def("scatter", [](Array &target, const Array &source, const Index &index, const Mask &mask) {
        scatter(target, source, index, mask);
    }, "target"_a, "source"_a, "index"_a, "mask"_a = Mask(true));
    m.def("scatter_add", [](Array &target, const Array &source, const Index &index, const Mask &mask) {
        scatter_add(target, source, index, mask);
    }, "target"_a, "source"_a, "index"_a, "mask"_a = Mask(true));
}

/// Conversions between the array classes of one module (Float32(UInt32) etc.)
template <typename Dst, typename Src> void bind_cast(py::class_<Dst> &cl) {
    cl.def(py::init([](const Src &s) { return Dst(s); }));
}

/// Static vectors of device arrays: Vector{1,2,3,4}{m,i,u,f,d} (cuda_1d.cpp ... cuda_4d.cpp / cuda_autodiff_*d.cpp of the
/// reference bind the same family through bind<>, src/python/common.h:338-998).  The operator set follows the element
/// type: masks get the logical operators and all / any, integers + - * and the bit operators, floats the full arithmetic
/// and the geometric helpers.
template <typename Value, size_t N> py::class_<Array<Value, N>> bind_vector(py::module_ &m, const char *name) {
    using Vec = Array<Value, N>;
    using Scalar = scalar_t<Value>;
    using Mask = mask_t<Value>;
    using VecMask = mask_t<Vec>;
    using UInt32 = uint32_array_t<Value>;
    constexpr bool IsMaskV = is_mask_v<Value>, IsFloatV = std::is_floating_point_v<Scalar>,
                   IsIntV = std::is_integral_v<Scalar> && !IsMaskV;
    py::class_<Vec> cl(m, name);
    cl.def(py::init<>())
      .def(py::init<const Vec &>())
      .def(py::init<Scalar>())
      .def(py::init<const Value &>())
      .def("__len__", [](const Vec &) { return N; })
      .def("__getitem__", [](const Vec &v, size_t i) { if (i >= N) throw py::index_error(); return v.coeff(i); })
      .def("__setitem__", [](Vec &v, size_t i, const Value &x) { if (i >= N) throw py::index_error(); v.coeff(i) = x; })
      .def("__setitem__", [](Vec &v, const Mask &mk, const Vec &x) { masked(v, mk) = x; })
      .def("__repr__", [](const Vec &v) {
          std::string s = "[";
          for (size_t i = 0; i < N; ++i) s += array_repr(v.coeff(i)) + (i + 1 < N ? ",\n " : "]");
          return s;
      });
    if constexpr (N == 2) cl.def(py::init<const Value &, const Value &>());
    if constexpr (N == 3) cl.def(py::init<const Value &, const Value &, const Value &>());
    if constexpr (N == 4) cl.def(py::init<const Value &, const Value &, const Value &, const Value &>());
    cl.def_property("x", [](const Vec &v) { return v.x(); }, [](Vec &v, const Value &x) { v.x() = x; });
    if constexpr (N >= 2) cl.def_property("y", [](const Vec &v) { return v.y(); }, [](Vec &v, const Value &x) { v.y() = x; });
    if constexpr (N >= 3) cl.def_property("z", [](const Vec &v) { return v.z(); }, [](Vec &v, const Value &x) { v.z() = x; });
    if constexpr (N >= 4) cl.def_property("w", [](const Vec &v) { return v.w(); }, [](Vec &v, const Value &x) { v.w() = x; });

    if constexpr (IsMaskV) {
        cl.def("__and__", [](const Vec &a, const Vec &b) { return Vec(a & b); })
          .def("__or__", [](const Vec &a, const Vec &b) { return Vec(a | b); })
          .def("__xor__", [](const Vec &a, const Vec &b) { return Vec(a ^ b); })
          .def("__invert__", [](const Vec &a) { return Vec(!a); });
        m.def("all", [](const Vec &a) { return all(a); });          // over the components -> a mask array (cuda semantics)
        m.def("any", [](const Vec &a) { return any(a); });
        m.def("all_nested", [](const Vec &a) { return all_nested(a); });
        m.def("any_nested", [](const Vec &a) { return any_nested(a); });
        m.def("none_nested", [](const Vec &a) { return none_nested(a); });
    } else {
        cl.def(py::self + py::self).def(py::self - py::self).def(py::self * py::self).def(-py::self)
          .def("__mul__", [](const Vec &a, const Value &b) { return Vec(a * b); })
          .def("__rmul__", [](const Vec &a, const Value &b) { return Vec(a * b); })
          .def("__mul__", [](const Vec &a, Scalar b) { return Vec(a * b); })
          .def("__rmul__", [](const Vec &a, Scalar b) { return Vec(a * b); })
          .def("__add__", [](const Vec &a, Scalar b) { return Vec(a + b); })
          .def("__sub__", [](const Vec &a, Scalar b) { return Vec(a - b); })
          .def("__lt__", [](const Vec &a, const Vec &b) { return VecMask(a < b); })
          .def("__le__", [](const Vec &a, const Vec &b) { return VecMask(a <= b); })
          .def("__gt__", [](const Vec &a, const Vec &b) { return VecMask(a > b); })
          .def("__ge__", [](const Vec &a, const Vec &b) { return VecMask(a >= b); });
        m.def("eq", [](const Vec &a, const Vec &b) { return VecMask(eq(a, b)); });
        m.def("neq", [](const Vec &a, const Vec &b) { return VecMask(neq(a, b)); });
        m.def("hsum_nested", [](const Vec &a) { return hsum(hsum(a)); });
        m.def("hprod_nested", [](const Vec &a) { return hprod(hprod(a)); });
        m.def("hmin_nested", [](const Vec &a) { return hmin(hmin(a)); });
        m.def("hmax_nested", [](const Vec &a) { return hmax(hmax(a)); });
        m.def("hsum", [](const Vec &a) { return hsum(a); });
        m.def("hprod", [](const Vec &a) { return hprod(a); });
        m.def("hmin", [](const Vec &a) { return hmin(a); });
        m.def("hmax", [](const Vec &a) { return hmax(a); });
        m.def("abs", [](const Vec &a) { return abs(a); });
        m.def("min", [](const Vec &a, const Vec &b) { return min(a, b); });
        m.def("max", [](const Vec &a, const Vec &b) { return max(a, b); });
        m.def("fmadd", [](const Vec &a, const Vec &b, const Vec &c) { return fmadd(a, b, c); });
        m.def("dot", [](const Vec &a, const Vec &b) { return dot(a, b); });
        m.def("scatter_add", [](Vec &target, const Vec &source, const UInt32 &index, const Mask &mask) { scatter_add(target, source, index, mask); },
              "target"_a, "source"_a, "index"_a, "mask"_a = Mask(true));
    }
    if constexpr (IsIntV) {
        cl.def("__and__", [](const Vec &a, const Vec &b) { return Vec(a & b); })
          .def("__or__", [](const Vec &a, const Vec &b) { return Vec(a | b); })
          .def("__xor__", [](const Vec &a, const Vec &b) { return Vec(a ^ b); })
          .def("__lshift__", [](const Vec &a, const Vec &b) { return Vec(a << b); })
          .def("__rshift__", [](const Vec &a, const Vec &b) { return Vec(a >> b); })
          .def("__floordiv__", [](const Vec &a, const Vec &b) { return Vec(a / b); })
          .def("__mod__", [](const Vec &a, const Vec &b) { return Vec(a % b); });
        if constexpr (std::is_unsigned_v<Scalar> && N >= 2 && N <= 4) {
            // Morton / Z-order codes (include/enoki/morton.h)
            m.def("morton_encode", [](const Vec &coords) { return morton_encode(coords); });
            cl.def_static("morton_decode", [](const Value &code) { return morton_decode<Vec>(code); });
        }
    }
    if constexpr (IsFloatV) {
        cl.def(py::self / py::self)
          .def("__truediv__", [](const Vec &a, const Value &b) { return Vec(a / b); })
          .def("__truediv__", [](const Vec &a, Scalar b) { return Vec(a / b); });
        m.def("abs_dot", [](const Vec &a, const Vec &b) { return abs(dot(a, b)); });
        m.def("squared_norm", [](const Vec &a) { return squared_norm(a); });
        m.def("norm", [](const Vec &a) { return norm(a); });
        m.def("normalize", [](const Vec &a) { return normalize(a); });
        m.def("sqrt", [](const Vec &a) { return sqrt(a); });
        if constexpr (N == 3) m.def("cross", [](const Vec &a, const Vec &b) { return cross(a, b); });
        if constexpr (N == 3)
            m.def("sh_eval", [](const Vec &d, size_t order) {
                std::vector<Value> out((order + 1) * (order + 1));
                sh_eval(d, order, out.data());
                return out;
            }, "d"_a, "order"_a, "real spherical harmonics Y_l^m(d), index l * (l + 1) + m (include/enoki/sh.h)");
    }
    m.def("select", [](const Mask &mk, const Vec &t, const Vec &f) { return select(mk, t, f); });
    m.def("slices", [](const Vec &a) { return slices(a); });
    m.def("gather", [](const Vec &source, const UInt32 &index, const Mask &mask) { return gather<Vec>(source, index, mask); },
          "source"_a, "index"_a, "mask"_a = Mask(true));
    m.def("scatter", [](Vec &target, const Vec &source, const UInt32 &index, const Mask &mask) { scatter(target, source, index, mask); },
          "target"_a, "source"_a, "index"_a, "mask"_a = Mask(true));
    if constexpr (is_diff_array_v<Value> && IsFloatV) {
        m.def("set_requires_gradient", [](Vec &a, bool value) { set_requires_gradient(a, value); }, "array"_a, "value"_a = true);
        m.def("gradient", [](const Vec &a) { return gradient(a); });
        m.def("detach", [](const Vec &a) { return detach(a); });
    }
    return cl;
}

/// Conversions between the element flavours of one vector size (cuda_3d.cpp:10-33): Vector3f(Vector3i) ...
template <typename Dst, typename Src> void bind_vector_cast(py::class_<Dst> &cl) {
    cl.def(py::init([](const Src &s) { return Dst(s); }));
}

/// Vector0*: the degenerate size-0 members of the family (cuda_0d.cpp); nothing to compute, kept for API completeness
template <typename Tag> struct EmptyVector { };
template <typename Tag> void bind_vector0(py::module_ &m, const char *name) {
    py::class_<EmptyVector<Tag>>(m, name)
        .def(py::init<>())
        .def("__len__", [](const EmptyVector<Tag> &) { return 0; })
        .def("__repr__", [](const EmptyVector<Tag> &) { return std::string("[]"); });
}

/// The whole family for one module: `A<T>` maps an element type to the module's array type
template <template <typename> class A> void bind_vector_family(py::module_ &m) {
    bind_vector0<A<bool>>(m, "Vector0m"); bind_vector0<A<int32_t>>(m, "Vector0i"); bind_vector0<A<uint32_t>>(m, "Vector0u");
    bind_vector0<A<float>>(m, "Vector0f"); bind_vector0<A<double>>(m, "Vector0d");
#define ENOKI_BIND_VECTORS(N)                                                                                      \
    {                                                                                                              \
        auto vm = bind_vector<A<bool>, N>(m, "Vector" #N "m");                                                     \
        auto vi = bind_vector<A<int32_t>, N>(m, "Vector" #N "i");                                                  \
        auto vu = bind_vector<A<uint32_t>, N>(m, "Vector" #N "u");                                                 \
        auto vf = bind_vector<A<float>, N>(m, "Vector" #N "f");                                                    \
        auto vd = bind_vector<A<double>, N>(m, "Vector" #N "d");                                                   \
        (void) vm;                                                                                                 \
        bind_vector_cast<Array<A<float>, N>, Array<A<double>, N>>(vf); bind_vector_cast<Array<A<float>, N>, Array<A<int32_t>, N>>(vf);   \
        bind_vector_cast<Array<A<float>, N>, Array<A<uint32_t>, N>>(vf);                                           \
        bind_vector_cast<Array<A<double>, N>, Array<A<float>, N>>(vd); bind_vector_cast<Array<A<double>, N>, Array<A<int32_t>, N>>(vd);  \
        bind_vector_cast<Array<A<double>, N>, Array<A<uint32_t>, N>>(vd);                                          \
        bind_vector_cast<Array<A<int32_t>, N>, Array<A<uint32_t>, N>>(vi); bind_vector_cast<Array<A<int32_t>, N>, Array<A<float>, N>>(vi); \
        bind_vector_cast<Array<A<int32_t>, N>, Array<A<double>, N>>(vi);                                           \
        bind_vector_cast<Array<A<uint32_t>, N>, Array<A<int32_t>, N>>(vu); bind_vector_cast<Array<A<uint32_t>, N>, Array<A<float>, N>>(vu); \
        bind_vector_cast<Array<A<uint32_t>, N>, Array<A<double>, N>>(vu);                                          \
    }
    ENOKI_BIND_VECTORS(1) ENOKI_BIND_VECTORS(2) ENOKI_BIND_VECTORS(3) ENOKI_BIND_VECTORS(4)
#undef ENOKI_BIND_VECTORS
}

/// Matrix<Value, N> (src/python/matrix.h of the reference): N x N entries in row-major order, products, transpose, ...
template <typename Value, size_t N> py::class_<Matrix<Value, N>> bind_matrix(py::module_ &m, const char *name) {
    using Mat = Matrix<Value, N>;
    using Vec = Array<Value, N>;
    py::class_<Mat> cl(m, name);
    cl.def(py::init<>())
      .def(py::init<const Mat &>())
      .def(py::init<const Value &>(), "diagonal matrix")
      .def(py::init([](const std::vector<Value> &rows) {
          if (rows.size() != N * N) throw py::value_error("expected N*N entries in row-major order");
          Mat r;
          for (size_t i = 0; i < N; ++i)
              for (size_t j = 0; j < N; ++j)
                  r(i, j) = rows[i * N + j];
          return r;
      }), "entries"_a, "N*N entries in row-major order")
      .def_static("identity", [](size_t size) { return identity<Mat>(size); }, "size"_a = 1)
      .def_static("from_cols", [](const std::vector<Vec> &cols) {
          if (cols.size() != N) throw py::value_error("expected N columns");
          Mat r;
          for (size_t j = 0; j < N; ++j) r.col(j) = cols[j];
          return r;
      })
      .def("__getitem__", [](const Mat &a, std::pair<size_t, size_t> ij) {
          if (ij.first >= N || ij.second >= N) throw py::index_error();
          return a(ij.first, ij.second);
      })
      .def("__setitem__", [](Mat &a, std::pair<size_t, size_t> ij, const Value &v) {
          if (ij.first >= N || ij.second >= N) throw py::index_error();
          a(ij.first, ij.second) = v;
      })
      .def("col", [](const Mat &a, size_t j) { if (j >= N) throw py::index_error(); return Vec(a.col(j)); })
      .def("row", [](const Mat &a, size_t i) { if (i >= N) throw py::index_error(); return a.row(i); })
      .def("__matmul__", [](const Mat &a, const Mat &b) { return Mat(a * b); })
      .def("__matmul__", [](const Mat &a, const Vec &v) { return Vec(a * v); })
      .def("__mul__", [](const Mat &a, const Mat &b) { return Mat(a * b); })
      .def("__mul__", [](const Mat &a, const Vec &v) { return Vec(a * v); })
      .def("__mul__", [](const Mat &a, const Value &s) { return Mat(a * s); })
      .def("__rmul__", [](const Mat &a, const Value &s) { return Mat(s * a); })
      .def("__add__", [](const Mat &a, const Mat &b) { return Mat(a + b); })
      .def("__sub__", [](const Mat &a, const Mat &b) { return Mat(a - b); });
    // homogeneous transformations (include/enoki/transform.h; reference src/python/common.h binds them as static methods)
    if constexpr (N == 4) {
        using Vec3 = Array<Value, 3>;
        cl.def_static("translate", [](const Vec3 &v) { return translate<Mat>(v); })
          .def_static("scale", [](const Vec3 &v) { return scale<Mat>(v); })
          .def_static("rotate", [](const Vec3 &axis, const Value &angle) { return rotate<Mat>(axis, angle); }, "axis"_a, "angle"_a)
          .def_static("perspective", [](const Value &fov, const Value &near_, const Value &far_, const Value &aspect) {
              return perspective<Mat>(fov, near_, far_, aspect); }, "fov"_a, "near"_a, "far"_a, "aspect"_a = Value(scalar_t<Value>(1)))
          .def_static("frustum", [](const Value &l, const Value &r, const Value &b, const Value &t, const Value &n, const Value &f) {
              return frustum<Mat>(l, r, b, t, n, f); }, "left"_a, "right"_a, "bottom"_a, "top"_a, "near"_a, "far"_a)
          .def_static("ortho", [](const Value &l, const Value &r, const Value &b, const Value &t, const Value &n, const Value &f) {
              return ortho<Mat>(l, r, b, t, n, f); }, "left"_a, "right"_a, "bottom"_a, "top"_a, "near"_a, "far"_a)
          .def_static("look_at", [](const Vec3 &origin, const Vec3 &target, const Vec3 &up) { return look_at<Mat>(origin, target, up); },
                      "origin"_a, "target"_a, "up"_a);
    } else if constexpr (N == 3) {
        cl.def_static("rotate", [](const Value &angle) { return rotate<Mat>(angle); }, "angle"_a);
    }
    if constexpr (N == 3 || N == 4)
        m.def("polar_decomp", [](const Mat &a, size_t it) { return polar_decomp(a, it); }, "a"_a, "it"_a = 10,
              "A = Q P: (orthogonal Q, symmetric P) by the scaled Newton iteration (matrix.h)");
    if constexpr (N == 4) {
        using Vec3 = Array<Value, 3>;
        using Mat3 = Matrix<Value, 3>;
        m.def("transform_decompose", [](const Mat &a, size_t it) { return transform_decompose(a, it); }, "a"_a, "it"_a = 10,
              "affine 4x4 -> (scale / shear 3x3, rotation quaternion, translation)");
        m.def("transform_compose", [](const Mat3 &s, const Quaternion<Value> &q, const Vec3 &t) { return transform_compose(s, q, t); });
        m.def("transform_compose_inverse", [](const Mat3 &s, const Quaternion<Value> &q, const Vec3 &t) { return transform_compose_inverse(s, q, t); });
    }
    m.def("transpose", [](const Mat &a) { return Mat(transpose(a)); });
    cl.def_property_readonly("T", [](const Mat &a) { return Mat(transpose(a)); });
    m.def("trace", [](const Mat &a) { return trace(a); });
    m.def("frob", [](const Mat &a) { return frob(a); });
    m.def("diag", [](const Mat &a) { return Vec(diag(a)); });
    if constexpr (N >= 2 && N <= 4) {
        m.def("det", [](const Mat &a) { return det(a); });
        m.def("inverse", [](const Mat &a) { return Mat(inverse(a)); });
        m.def("inverse_transpose", [](const Mat &a) { return Mat(transpose(inverse(a))); });
    }
    return cl;
}

/// Complex<Value> (src/python/complex.h of the reference)
template <typename Value> py::class_<Complex<Value>> bind_complex(py::module_ &m, const char *name) {
    using C = Complex<Value>;
    py::class_<C> cl(m, name);
    cl.def(py::init<>())
      .def(py::init<const C &>())
      .def(py::init<const Value &>(), "real"_a)
      .def(py::init<const Value &, const Value &>(), "real"_a, "imag"_a)
      .def_property("real", [](const C &z) { return real(z); }, [](C &z, const Value &v) { z.coeff(0) = v; })
      .def_property("imag", [](const C &z) { return imag(z); }, [](C &z, const Value &v) { z.coeff(1) = v; })
      .def("__getitem__", [](const C &z, size_t i) { if (i >= 2) throw py::index_error(); return z.coeff(i); })
      .def("__len__", [](const C &) { return 2; })
      .def("__add__", [](const C &a, const C &b) { return C(a + b); })
      .def("__sub__", [](const C &a, const C &b) { return C(a - b); })
      .def("__neg__", [](const C &a) { return C(-a); })
      .def("__mul__", [](const C &a, const C &b) { return C(a * b); })
      .def("__mul__", [](const C &a, const Value &b) { return C(a * b); })
      .def("__rmul__", [](const C &a, const Value &b) { return C(b * a); })
      .def("__truediv__", [](const C &a, const C &b) { return C(a / b); })
      .def("__truediv__", [](const C &a, const Value &b) { return C(a / b); });
    m.def("real", [](const C &z) { return real(z); });
    m.def("imag", [](const C &z) { return imag(z); });
    m.def("conj", [](const C &z) { return conj(z); });
    m.def("squared_norm", [](const C &z) { return squared_norm(z); });
    m.def("abs", [](const C &z) { return abs(z); });
    m.def("arg", [](const C &z) { return arg(z); });
    m.def("rcp", [](const C &z) { return rcp(z); });
    m.def("exp", [](const C &z) { return exp(z); });
    m.def("log", [](const C &z) { return log(z); });
    m.def("sqrt", [](const C &z) { return sqrt(z); });
    m.def("pow", [](const C &a, const C &b) { return pow(a, b); });
    m.def("sin", [](const C &z) { return sin(z); });
    m.def("cos", [](const C &z) { return cos(z); });
    m.def("tan", [](const C &z) { return tan(z); });
    m.def("sinh", [](const C &z) { return sinh(z); });
    m.def("cosh", [](const C &z) { return cosh(z); });
    m.def("tanh", [](const C &z) { return tanh(z); });
    m.def("asin", [](const C &z) { return asin(z); });
    m.def("acos", [](const C &z) { return acos(z); });
    m.def("atan", [](const C &z) { return atan(z); });
    m.def("asinh", [](const C &z) { return asinh(z); });
    m.def("acosh", [](const C &z) { return acosh(z); });
    m.def("atanh", [](const C &z) { return atanh(z); });
    return cl;
}

/// Quaternion<Value> (src/python/quat.h of the reference)
template <typename Value> py::class_<Quaternion<Value>> bind_quaternion(py::module_ &m, const char *name) {
    using Q = Quaternion<Value>;
    using Scalar = scalar_t<Value>;
    using Vector3 = Array<Value, 3>;
    py::class_<Q> cl(m, name);
    cl.def(py::init<>())
      .def(py::init<const Q &>())
      .def(py::init<const Value &>(), "w"_a)
      .def(py::init<const Value &, const Value &, const Value &, const Value &>(), "x"_a, "y"_a, "z"_a, "w"_a)
      .def("__getitem__", [](const Q &q, size_t i) { if (i >= 4) throw py::index_error(); return q.coeff(i); })
      .def("__setitem__", [](Q &q, size_t i, const Value &v) { if (i >= 4) throw py::index_error(); q.coeff(i) = v; })
      .def("__len__", [](const Q &) { return 4; })
      .def("__add__", [](const Q &a, const Q &b) { return Q(a + b); })
      .def("__sub__", [](const Q &a, const Q &b) { return Q(a - b); })
      .def("__neg__", [](const Q &a) { return Q(-a); })
      .def("__mul__", [](const Q &a, const Q &b) { return Q(a * b); })
      .def("__mul__", [](const Q &a, const Value &b) { return Q(a * b); })
      .def("__truediv__", [](const Q &a, const Q &b) { return Q(a / b); })
      .def("__truediv__", [](const Q &a, const Value &b) { return Q(a / b); })
      .def_static("identity", [](size_t size) { return identity<Q>(size); }, "size"_a = 1)
      .def_static("zero", [](size_t size) { Value z = zero<Value>(size); return Q(z, z, z, z); }, "size"_a = 1)
      .def_static("full", [](Scalar v, size_t size) { Value f = full<Value>(v, size); return Q(f, f, f, f); }, "value"_a, "size"_a = 1);
    cl.def_property("x", [](const Q &q) { return q.x(); }, [](Q &q, const Value &v) { q.x() = v; });
    cl.def_property("y", [](const Q &q) { return q.y(); }, [](Q &q, const Value &v) { q.y() = v; });
    cl.def_property("z", [](const Q &q) { return q.z(); }, [](Q &q, const Value &v) { q.z() = v; });
    cl.def_property("w", [](const Q &q) { return q.w(); }, [](Q &q, const Value &v) { q.w() = v; });
    m.def("real", [](const Q &q) { return real(q); });
    m.def("imag", [](const Q &q) { return imag(q); });
    m.def("conj", [](const Q &q) { return conj(q); });
    m.def("norm", [](const Q &q) { return norm(q); });
    m.def("squared_norm", [](const Q &q) { return squared_norm(q); });
    m.def("rcp", [](const Q &q) { return rcp(q); });
    m.def("normalize", [](const Q &q) { return normalize(q); });
    m.def("dot", [](const Q &a, const Q &b) { return dot(a, b); });
    m.def("abs", [](const Q &q) { return abs(q); });
    m.def("sqrt", [](const Q &q) { return sqrt(q); });
    m.def("exp", [](const Q &q) { return exp(q); });
    m.def("log", [](const Q &q) { return log(q); });
    m.def("pow", [](const Q &a, const Q &b) { return pow(a, b); });
    m.def("slerp", [](const Q &a, const Q &b, const Value &t) { return slerp(a, b, t); }, "a"_a, "b"_a, "t"_a);
    m.def("quat_to_euler", [](const Q &q) { return quat_to_euler<Vector3>(q); });
    m.def("quat_to_matrix", [](const Q &q) { return quat_to_matrix<Matrix<Value, 4>>(q); });
    m.def("quat_to_matrix3", [](const Q &q) { return quat_to_matrix<Matrix<Value, 3>>(q); });
    m.def("matrix_to_quat", [](const Matrix<Value, 4> &mat) { return matrix_to_quat(mat); });
    m.def("matrix_to_quat", [](const Matrix<Value, 3> &mat) { return matrix_to_quat(mat); });
    m.def("rotate", [](const Vector3 &axis, const Value &angle) { return rotate<Q>(axis, angle); }, "axis"_a, "angle"_a);
    return cl;
}

inline void bind_runtime(py::module_ &m) {
    m.def("hip_eval", []() { hip_eval(); }, "no-op: the backend is eager (kept for cuda_eval() call sites)");
    m.def("hip_sync", []() { py::gil_scoped_release r; hip_sync(); });
    m.def("hip_whos", []() { return hip_whos(); });
    m.def("hip_malloc_trim", []() { hip_malloc_trim(); });
    m.def("hip_set_log_level", [](uint32_t l) { ek_hip_set_log_level(l); });
    m.def("hip_log_level", []() { return ek_hip_log_level(); });
    m.def("hip_mem_get_info", []() {
        size_t free_bytes = 0, total_bytes = 0;
        detail::hip_check(ek_hip_mem_get_info(&free_bytes, &total_bytes), "hip_mem_get_info");
        return std::make_pair(free_bytes, total_bytes);
    }, "(free, total) bytes of device memory (cuda_mem_get_info)");
    m.def("hip_init", [](int device) { detail::hip_check(ek_hip_init(device), "hip_init"); }, "device"_a = -1);
    m.def("hip_device", []() { return ek_hip_device(); });
    m.def("hip_stream", []() { return (uintptr_t) ek_hip_stream(); });
    m.def("hip_set_stream", [](uintptr_t s) { detail::hip_check(ek_hip_set_stream((void *) s), "hip_set_stream"); });
    m.def("hip_launch_count", []() { return ek_hip_launch_count(); });
    // step graphs: capture once, replay without host work (ek_hip_graph_*)
    m.def("hip_graph_begin", []() { hip_graph_begin(); },
          "start capturing a step graph; arrays that are still unevaluated (deferred gathers / unary results) are evaluated first");
    m.def("hip_graph_end", []() { return (uintptr_t) hip_graph_end(); },
          "ends the capture (arrays that are still unevaluated are evaluated INSIDE the graph first, so every replay refreshes "
          "them); returns a graph handle for hip_graph_launch / hip_graph_destroy");
    m.def("hip_graph_launch", [](uintptr_t g) { detail::hip_check(ek_hip_graph_launch((ek_hip_graph *) g), "hip_graph_launch"); });
    m.def("hip_graph_launch_count", [](uintptr_t g) { return ek_hip_graph_launch_count((const ek_hip_graph *) g); });
    m.def("hip_graph_destroy", [](uintptr_t g) { detail::hip_check(ek_hip_graph_destroy((ek_hip_graph *) g), "hip_graph_destroy"); });
    m.def("hip_set_defer_gather", [](bool v) { hip_set_defer_gather(v); },
          "large gathers from small tables stay deferred until consumed (fused into the consuming add/sub/mul/fma)");
    m.def("hip_set_defer", [](bool v) { hip_set_defer(v); }, "deferred evaluation of gathers and fusable unary ops on / off");
    m.def("hip_defer", []() { return hip_defer(); });
    m.def("hip_set_scatter_aliasing", [](bool v) { hip_set_scatter_aliasing(v); },
          "True: scatter / scatter_add write IN PLACE through shared handles, like copies of a CUDAArray that alias one variable "
          "(cuda.h:224-226); False (default): copy on write");
    m.def("hip_scatter_aliasing", []() { return hip_scatter_aliasing(); });
    m.def("hip_defer_gather", []() { return hip_defer_gather(); });
    m.def("hip_profile_begin", []() { detail::hip_check(ek_hip_profile_begin(), "hip_profile_begin"); });
    m.def("hip_profile_end", []() {
        char *r = ek_hip_profile_end();
        std::string s(r ? r : "[]");
        free(r);
        return s;
    }, "JSON list of {kernel, launches, total_ms, bytes, elements} since hip_profile_begin()");
    m.def("hip_concat_f32", [](uintptr_t out, const std::vector<std::pair<uintptr_t, size_t>> &parts) {
        std::vector<const void *> srcs;
        std::vector<size_t> sizes;
        for (const auto &p : parts) { srcs.push_back((const void *) p.first); sizes.push_back(p.second); }
        detail::hip_check(ek_hip_concat(EK_F32, (void *) out, (int) parts.size(), srcs.data(), sizes.data()), "hip_concat_f32");
    }, "out"_a, "parts"_a, "out = parts[0] | parts[1] | ... ((device pointer, element count) pairs of float32 arrays), one launch");
    m.def("hip_concat_rows_f32", [](uintptr_t out, size_t rows, const std::vector<std::pair<uintptr_t, size_t>> &parts) {
        std::vector<const void *> srcs;
        std::vector<size_t> sizes;
        for (const auto &p : parts) { srcs.push_back((const void *) p.first); sizes.push_back(p.second); }
        detail::hip_check(ek_hip_concat_rows(EK_F32, (void *) out, rows, (int) parts.size(), srcs.data(), sizes.data()), "hip_concat_rows_f32");
    }, "out"_a, "rows"_a, "parts"_a, "out[r] = parts[0][r] | parts[1][r] | ... with every part seen as [rows, size / rows]: the staging "
       "layout of a reduce-scatter, one launch");
    m.def("hip_set_tuning", [](const char *k, int v) { detail::hip_check(ek_hip_set_tuning(k, v), "hip_set_tuning"); });
}
