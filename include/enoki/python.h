/*
    enoki/python.h -- pybind11 support for HOST arrays (reference: include/enoki/python.h)

    A type caster between static arrays of arithmetic scalars -- Array<float, 3>, Array<Array<double, 4>, 4>, masks -- and
    NumPy arrays, for extension modules that take or return such values.  As in the reference the NumPy axes run from the
    INNERMOST array dimension to the outermost one: Array<Array<float, 4>, 3> <-> ndarray of shape (4, 3), so that
    `a[i]` on the Python side is one slice (x_i, y_i, z_i).  Anything NumPy can convert (lists, tuples, other dtypes with
    `convert`) is accepted; `None` is not.

    Device arrays (HIPArray<T>, DiffArray<...>, Array<HIPArray<T>, N>, Matrix, ...) do not go through a caster: they are
    classes registered by the modules enoki_amd.hip / enoki_amd.hip_autodiff, and an extension module that includes the
    array headers takes and returns them directly (tests/cpp/user_ext/user_ext.cpp) -- the same split as in the reference,
    whose caster excludes CUDA arrays (python.h:48-50).
*/
#pragma once

#include <enoki/array.h>

#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include <vector>

namespace enoki::detail {
    /// shape with the innermost dimension first, e.g. {4, 3} for Array<Array<float, 4>, 3>
    template <typename T> inline void numpy_shape(std::vector<pybind11::ssize_t> &out) {
        if constexpr (is_array_v<T>) {
            numpy_shape<value_t<T>>(out);
            out.push_back((pybind11::ssize_t) std::decay_t<T>::Size);
        }
    }
    /// visit every scalar with its NumPy multi-index (innermost dimension first)
    template <typename T, typename F> inline void numpy_visit(T &a, std::vector<pybind11::ssize_t> &index, size_t level, F &&fn) {
        if constexpr (is_array_v<std::decay_t<T>>) {
            for (size_t i = 0; i < std::decay_t<T>::Size; ++i) {
                index[level] = (pybind11::ssize_t) i;
                numpy_visit(a.coeff(i), index, level - 1, fn);
            }
        } else {
            fn(a, index);
        }
    }
}

namespace pybind11::detail {

template <typename Value>
struct type_caster<Value, std::enable_if_t<enoki::is_array_v<Value> && !enoki::is_dynamic_v<Value> &&
                                           std::is_arithmetic_v<enoki::scalar_t<Value>>>> {
    using Scalar = enoki::scalar_t<Value>;
    PYBIND11_TYPE_CASTER(Value, const_name("numpy.ndarray"));

    bool load(handle src, bool convert) {
        if (src.is_none()) return false;
        if (!convert && !isinstance<array_t<Scalar>>(src)) return false;
        auto arr = array_t<Scalar, array::c_style | array::forcecast>::ensure(src);
        if (!arr) { PyErr_Clear(); return false; }
        std::vector<ssize_t> shape;
        enoki::detail::numpy_shape<Value>(shape);
        if ((size_t) arr.ndim() != shape.size()) return false;
        for (size_t d = 0; d < shape.size(); ++d)
            if (arr.shape((ssize_t) d) != shape[d]) return false;
        const Scalar *data = arr.data();
        std::vector<ssize_t> strides(shape.size());
        for (size_t d = 0; d < shape.size(); ++d) strides[d] = arr.strides((ssize_t) d) / (ssize_t) sizeof(Scalar);
        std::vector<ssize_t> index(shape.size(), 0);
        enoki::detail::numpy_visit(value, index, shape.size() - 1, [&](Scalar &entry, const std::vector<ssize_t> &at) {
            ssize_t offset = 0;
            for (size_t d = 0; d < at.size(); ++d) offset += at[d] * strides[d];
            entry = data[offset];
        });
        return true;
    }

    static handle cast(const Value &v, return_value_policy, handle) {
        std::vector<ssize_t> shape;
        enoki::detail::numpy_shape<Value>(shape);
        array_t<Scalar> out(shape);
        Scalar *data = out.mutable_data();
        std::vector<ssize_t> strides(shape.size());
        for (size_t d = 0; d < shape.size(); ++d) strides[d] = out.strides((ssize_t) d) / (ssize_t) sizeof(Scalar);
        std::vector<ssize_t> index(shape.size(), 0);
        enoki::detail::numpy_visit(v, index, shape.size() - 1, [&](const Scalar &entry, const std::vector<ssize_t> &at) {
            ssize_t offset = 0;
            for (size_t d = 0; d < at.size(); ++d) offset += at[d] * strides[d];
            data[offset] = entry;
        });
        return out.release();
    }
};

} // namespace pybind11::detail
