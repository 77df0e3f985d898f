"""A downstream pybind11 extension (tests/cpp/user_ext/user_ext.cpp) that includes only the array headers and binds its own
functions over HIPArray / DiffArray types: the classes registered by enoki_amd.hip / enoki_amd.hip_autodiff cross the
module boundary, as `enoki.cuda` types do for the reference's users (src/python/common.h registers them globally)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "cpp", "user_ext"))


def test_signatures_name_the_library_types():
    import enoki_amd.hip            # noqa: F401  (registers the types)
    import enoki_amd.hip_autodiff   # noqa: F401
    import user_ext
    assert "enoki_amd.hip.Float32" in user_ext.saxpy.__doc__
    assert "enoki_amd.hip.Vector3f" in user_ext.shade.__doc__ and "enoki_amd.hip_autodiff.Vector3f" in user_ext.shade.__doc__
    assert "enoki_amd.hip.UInt32" in user_ext.lookup.__doc__ and "enoki_amd.hip.Mask" in user_ext.lookup.__doc__


def test_host_static_arrays_travel_as_numpy():
    """enoki/python.h: Array<float, 3> etc. <-> NumPy, innermost dimension first (no GPU involved)"""
    import enoki_amd.hip            # noqa: F401
    import enoki_amd.hip_autodiff   # noqa: F401
    import user_ext
    d = np.array([1.0, -1.0, 0.5], np.float32); n = np.array([0.0, 1.0, 0.0], np.float32)
    r = user_ext.reflect(d, n)
    assert isinstance(r, np.ndarray) and r.dtype == np.float32 and np.array_equal(r, d - n * 2 * d.dot(n))
    assert np.array_equal(user_ext.reflect([1, -1, 0.5], (0, 1, 0)), r)          # lists / tuples / other dtypes convert
    o = user_ext.outer(d, np.array([2.0, 3.0], np.float32))
    assert o.shape == (3, 2) and np.array_equal(o, np.outer(d, [2.0, 3.0]).astype(np.float32))
    assert user_ext.trace(np.array([[1.0, 7.0], [9.0, 4.0]])) == 5.0
    assert user_ext.positive(d).tolist() == [True, False, True]
    with pytest.raises(TypeError):
        user_ext.reflect(np.zeros(4, np.float32), n)                               # wrong shape
    with pytest.raises(TypeError):
        user_ext.trace(np.zeros((2, 3)))                                           # wrong shape in the second axis
    with pytest.raises(TypeError):
        user_ext.reflect(None, n)                                                  # None is not an array
    assert user_ext.trace(np.array([[1, 2], [3, 4]], np.int32)) == 5.0             # other dtypes convert
    assert user_ext.trace(np.asfortranarray(np.array([[1.0, 2.0], [3.0, 4.0]]))) == 5.0      # any memory order
    strided = np.arange(12, dtype=np.float32)[::4]                                 # non-contiguous input
    assert np.array_equal(user_ext.reflect(strided, n), strided - n * 2 * strided.dot(n))


@pytest.mark.gpu
def test_downstream_functions_on_device_arrays():
    import enoki_amd.hip as ekc
    import enoki_amd.hip_autodiff as ekd
    import user_ext
    rng = np.random.default_rng(4)
    n = 100003
    x = rng.standard_normal(n).astype(np.float32); y = rng.standard_normal(n).astype(np.float32)
    got = user_ext.saxpy(2.5, ekc.Float32(x), ekc.Float32(y))
    assert isinstance(got, ekc.Float32)
    assert np.array_equal(got.numpy().view(np.uint32), ekc.fmadd(ekc.Float32(2.5), ekc.Float32(x), ekc.Float32(y)).numpy().view(np.uint32))
    assert user_ext.count_positive(ekc.Float32(x)) == int((x > 0).sum())

    nrm = [rng.standard_normal(n).astype(np.float32) for _ in range(3)]
    alb = rng.random(n).astype(np.float32)
    N = ekc.Vector3f(*[ekc.Float32(c) for c in nrm]); L = ekc.Vector3f(ekc.Float32(0.0), ekc.Float32(0.6), ekc.Float32(0.8))
    shaded = user_ext.shade(N, L, ekc.Float32(alb))
    expect = ekc.Float32(alb) * ekc.max(ekc.dot(ekc.normalize(N), L), ekc.Float32(0.0))
    assert np.array_equal(shaded.numpy().view(np.uint32), expect.numpy().view(np.uint32))

    # the differentiable overload: d shade / d albedo = max(dot(n, l), 0)
    ND = ekd.Vector3f(*[ekd.Float32(c) for c in nrm]); LD = ekd.Vector3f(ekd.Float32(0.0), ekd.Float32(0.6), ekd.Float32(0.8))
    albedo = ekd.Float32(alb)
    ekd.set_requires_gradient(albedo)
    out = user_ext.shade(ND, LD, albedo)
    assert isinstance(out, ekd.Float32)
    ekd.backward(ekd.hsum(out))
    cosine = ekc.max(ekc.dot(ekc.normalize(N), L), ekc.Float32(0.0)).numpy()
    assert np.array_equal(ekd.gradient(albedo).numpy().view(np.uint32), cosine.view(np.uint32))

    k = 5000
    table = [rng.standard_normal(k).astype(np.float32) for _ in range(3)]
    idx = rng.integers(0, k, n).astype(np.uint32); msk = rng.random(n) < 0.5
    r = user_ext.lookup(ekc.Vector3f(*[ekc.Float32(t) for t in table]), ekc.UInt32(idx), ekc.Mask(msk))
    for c, got in enumerate((r.x, r.y, r.z)):
        assert np.array_equal(got.numpy(), np.where(msk, table[c][idx], np.float32(0)))
