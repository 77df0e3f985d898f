"""Matrix<Value, N> over device arrays (include/enoki/matrix.h; reference include/enoki/matrix.h:20-318,
src/python/matrix.h).  tests/golden/matrix.npz comes from the reference build (oracle/ref_driver.cpp:ref_matrix):
products, trace and frob follow the reference's fmadd chains and must agree bit for bit; det / inverse of the
2 x 2 and 3 x 3 cases contain one rcp() (parity class C) and are compared within a few ulp."""
import os

import numpy as np
import pytest

from conftest import bits_equal

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(N):
    z = np.load(os.path.join(GOLDEN, "matrix.npz"))
    return {k[len(f"m{N}_"):]: z[k] for k in z.files if k.startswith(f"m{N}_")}


@pytest.mark.parametrize("N", [2, 3, 4])
def test_restated_operation_order_matches_reference(oracle, N):
    """CPU: column c_j of a * b = a.col(0) * b(0, j), then fmadd(a.col(i), b(i, j), .) -- the order matrix.h uses"""
    g = golden(N)
    a, b, v = g["a"], g["b"], g["v"]
    A = lambda i, j: a[i * N + j]
    B = lambda i, j: b[i * N + j]
    for i in range(N):
        acc = A(i, 0) * v[0]
        for k in range(1, N):
            acc = oracle.ternary("fmadd", A(i, k), v[k], acc)
        assert bits_equal(acc, g["mv"][i])
        for j in range(N):
            acc = A(i, 0) * B(0, j)
            for k in range(1, N):
                acc = oracle.ternary("fmadd", A(i, k), B(k, j), acc)
            assert bits_equal(acc, g["mm"][i * N + j])
    tr = A(0, 0)
    for i in range(1, N):
        tr = tr + A(i, i)
    assert bits_equal(tr, g["trace"])


def ulp_diff(x, y):
    xi = x.view(np.int32).astype(np.int64); yi = y.view(np.int32).astype(np.int64)
    return np.abs(xi - yi).max()


@pytest.mark.gpu
@pytest.mark.parametrize("mod", ["hip", "hip_autodiff"])
@pytest.mark.parametrize("N", [2, 3, 4])
def test_matrix_ops_match_reference(mod, N):
    import importlib
    ek = importlib.import_module(f"enoki_amd.{mod}")
    g = golden(N)
    M = getattr(ek, f"Matrix{N}f"); V = getattr(ek, f"Vector{N}f")
    a = M([ek.Float32(r) for r in g["a"]]); b = M([ek.Float32(r) for r in g["b"]])
    v = V(*[ek.Float32(r) for r in g["v"]])
    num = lambda x: (ek.detach(x) if mod == "hip_autodiff" else x).numpy()
    c = a @ b
    w = a @ v
    for i in range(N):
        assert bits_equal(num(w[i]), g["mv"][i])
        for j in range(N):
            assert bits_equal(num(c[i, j]), g["mm"][i * N + j]), (i, j)
            assert bits_equal(num(ek.transpose(a)[j, i]), g["a"][i * N + j])
    assert bits_equal(num(ek.trace(a)), g["trace"])
    # frob: fmadd chain over the columns is exact, the horizontal sum over N entries runs in index order
    assert np.allclose(num(ek.frob(a)), g["frob"], rtol=1e-6)
    assert bits_equal(num(ek.diag(a)[N - 1]), g["a"][N * N - 1])
    if N <= 4:
        # N = 2, 3: the reference's operation order (one rcp -> class C); N = 4: Laplace expansion over 2 x 2 minors
        # instead of the reference's shuffle formulation -> equal to rounding only
        if N <= 3:
            assert ulp_diff(num(ek.det(a)), g["det"]) <= 4
        else:
            assert np.allclose(num(ek.det(a)), g["det"], rtol=2e-5)
        ia = ek.inverse(a)
        for i in range(N):
            for j in range(N):
                assert np.allclose(num(ia[i, j]), g["inv"][i * N + j], rtol=4e-6 if N <= 3 else 1e-4, atol=1e-7 if N <= 3 else 2e-6), (i, j)
        # a * a^-1 = identity
        p = a @ ia
        for i in range(N):
            for j in range(N):
                assert np.allclose(num(p[i, j]), 1.0 if i == j else 0.0, atol=2e-6 if N <= 3 else 1e-5)
    ident = M.identity(5)
    assert np.array_equal(num(ident[0, 0]), np.ones(5, np.float32)) and np.array_equal(num(ident[0, N - 1]), np.zeros(5, np.float32))
    s = a * ek.Float32(2.0)
    assert bits_equal(num(s[1, 0]), g["a"][N] * np.float32(2))


@pytest.mark.gpu
def test_matrix_gradient():
    """d/dv hsum(M v) = column sums of M^T: the tape sees the fmadd chain like any other program"""
    import enoki_amd.hip_autodiff as ek
    g = golden(3)
    a = ek.Matrix3f([ek.Float32(r) for r in g["a"]])
    vs = [ek.Float32(r) for r in g["v"]]
    for x in vs:
        ek.set_requires_gradient(x)
    w = a @ ek.Vector3f(*vs)
    ek.backward(ek.hsum(w[0] + w[1] + w[2]))
    for k in range(3):
        want = g["a"][0 * 3 + k].astype(np.float64) + g["a"][1 * 3 + k] + g["a"][2 * 3 + k]
        assert np.allclose(ek.gradient(vs[k]).numpy(), want, rtol=1e-6)
