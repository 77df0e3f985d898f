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


@pytest.mark.extras
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


@pytest.mark.extras
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


# ---- transform.h: translate, scale, rotate, perspective, frustum, ortho, look_at --------------------------------------
TRANSFORMS = ["translate", "scale", "rotate4", "perspective", "frustum", "ortho", "look_at", "rotate3"]


def _close(a, b):
    return np.all(np.abs(a.astype(np.float64) - b) <= 2e-6 + 2e-6 * np.abs(b))


def test_transform_host_matches_golden():
    """include/enoki/transform.h on scalar entries (tests/cpp/transform_host.cpp) against the matrices of the reference build
    (tests/golden/transform.npz): translate / scale bit-exact, the rest to class C (the reference's rcp / rsqrt are
    rcpps / rsqrtps + Newton; the host's sincos is libm)"""
    import ctypes
    lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "libtransform_host.so"))
    z = np.load(os.path.join(GOLDEN, "transform.npz"))
    out = np.empty_like(z["out"])
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lib.transform_host(p(z["v"]), p(z["p"]), ctypes.c_size_t(z["v"].shape[1]), p(out))
    for k, name in enumerate(TRANSFORMS):
        if name in ("translate", "scale"):
            assert bits_equal(out[k], z["out"][k]), name
        else:
            assert _close(out[k], z["out"][k]), name
    # sanity of the conventions: a perspective matrix maps the near / far planes to -1 / +1, look_at moves the origin to 0
    n, f = z["p"][2].astype(np.float64), z["p"][3].astype(np.float64)
    P = z["out"][3].reshape(4, 4, -1).astype(np.float64)
    for depth, want in ((-n, -1.0), (-f, 1.0)):
        clip = P[2, 2] * depth + P[2, 3]
        w = P[3, 2] * depth
        assert np.allclose(clip / w, want, atol=1e-4)


@pytest.mark.extras
def test_transform_device_matches_golden():
    import enoki_amd.hip as ek
    z = np.load(os.path.join(GOLDEN, "transform.npz"))
    v = ek.Vector3f(*[ek.Float32(z["v"][i]) for i in range(3)])
    angle, fov, nr, fr, aspect = (ek.Float32(z["p"][i]) for i in range(5))
    one = ek.Float32(np.ones(z["v"].shape[1], np.float32))
    M = ek.Matrix4f
    mats = [M.translate(v), M.scale(v), M.rotate(ek.normalize(v), angle), M.perspective(fov, nr, fr, aspect),
            M.frustum(-aspect, aspect, -one, one, nr, fr), M.ortho(-aspect, aspect, -one, one, nr, fr),
            M.look_at(v, ek.Vector3f(*[v[i] * ek.Float32(0.25) + ek.Float32(1.0) for i in range(3)]),
                      ek.Vector3f(ek.Float32(0.0), ek.Float32(1.0), ek.Float32(0.0)))]
    n = z["v"].shape[1]
    for k, m in enumerate(mats):
        for i in range(4):
            for j in range(4):
                got = np.broadcast_to(m[i, j].numpy(), (n,))
                want = z["out"][k][i * 4 + j]
                if TRANSFORMS[k] in ("translate", "scale"):
                    assert bits_equal(np.ascontiguousarray(got), want), (TRANSFORMS[k], i, j)
                else:
                    assert _close(got, want), (TRANSFORMS[k], i, j)
    r = ek.Matrix3f.rotate(angle)
    for i in range(3):
        for j in range(3):
            assert _close(np.broadcast_to(r[i, j].numpy(), (n,)), z["out"][7][i * 3 + j]), ("rotate3", i, j)


def test_polar_decomposition_host_properties():
    """polar_decomp / transform_decompose / transform_compose(_inverse) on host scalars: tests/cpp/polar_host.cpp"""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "polar_host.bin")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.extras
def test_transform_decompose_device():
    """device arrays: compose(decompose(A)) = A and A * compose_inverse = I; Q of the polar decomposition against scipy"""
    import enoki_amd.hip as ek
    from scipy.linalg import polar
    rng = np.random.default_rng(4)
    n = 500
    A = np.tile(np.eye(4, dtype=np.float32)[:, :, None], (1, 1, n))
    A[:3, :, :] = rng.uniform(-2, 2, (3, 4, n)).astype(np.float32)
    keep = np.abs(np.linalg.det(np.moveaxis(A[:3, :3], 2, 0))) > 0.2
    A = A[:, :, keep]; n = A.shape[2]
    M = ek.Matrix4f([ek.Float32(np.ascontiguousarray(A[i, j])) for i in range(4) for j in range(4)])
    S, q, t = ek.transform_decompose(M)
    B = ek.transform_compose(S, q, t)
    I = B @ ek.transform_compose_inverse(S, q, t)
    for i in range(4):
        for j in range(4):
            assert np.allclose(np.broadcast_to(B[i, j].numpy(), (n,)), A[i, j], atol=2e-4), (i, j)
            assert np.allclose(np.broadcast_to(I[i, j].numpy(), (n,)), 1.0 if i == j else 0.0, atol=2e-4), (i, j)
    M3 = ek.Matrix3f([ek.Float32(np.ascontiguousarray(A[i, j])) for i in range(3) for j in range(3)])
    Q, P = ek.polar_decomp(M3)
    for s in range(0, n, 37):
        u, _ = polar(A[:3, :3, s].astype(np.float64))
        got = np.array([[Q[i, j].numpy()[s] for j in range(3)] for i in range(3)])
        assert np.allclose(got, u, atol=5e-4), s

