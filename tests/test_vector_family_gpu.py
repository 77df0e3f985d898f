"""The Vector{0..4}{m,i,u,f,d} family, Matrix*d, Complex2d and Quaternion4{f,d} of both python modules (the reference binds
them in src/python/cuda_{0..4}d.cpp, cuda_autodiff_{0..4}d.cpp, cuda_matrix.cpp, cuda_complex.cpp, quat.h)."""
import importlib
import os

import numpy as np
import pytest

from conftest import bits_equal

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NP = {"f": np.float32, "d": np.float64, "i": np.int32, "u": np.uint32}
ARR = {"f": "Float32", "d": "Float64", "i": "Int32", "u": "UInt32", "m": "Mask"}


@pytest.fixture(params=["hip", "hip_autodiff"])
def ek(request):
    m = importlib.import_module(f"enoki_amd.{request.param}")
    m.hip_init(0)
    return m


def _num(ek, x):
    return (ek.detach(x) if ek.__name__.endswith("autodiff") and hasattr(x, "numpy") is False else x).numpy()


def test_family_is_complete(ek):
    for n in range(5):
        for k in "miufd":
            assert hasattr(ek, f"Vector{n}{k}"), f"Vector{n}{k}"
    assert len(ek.Vector0f()) == 0 and repr(ek.Vector0m()) == "[]"
    for name in ("Matrix2d", "Matrix3d", "Matrix4d", "Complex2d", "Quaternion4f", "Quaternion4d"):
        assert hasattr(ek, name), name


@pytest.mark.parametrize("n", [1, 2, 3, 4])
@pytest.mark.parametrize("kind", ["i", "u", "f", "d"])
def test_vector_arithmetic_gather_scatter(ek, n, kind):
    rng = np.random.default_rng(n * 7 + ord(kind))
    dt = NP[kind]
    A = getattr(ek, ARR[kind]); V = getattr(ek, f"Vector{n}{kind}")
    size, K = 5000, 300
    comps = [(rng.integers(1, 50, size) if kind in "iu" else rng.standard_normal(size)).astype(dt) for _ in range(2 * n)]
    a, b = V(*[A(c) for c in comps[:n]]) if n > 1 else V(A(comps[0])), V(*[A(c) for c in comps[n:]]) if n > 1 else V(A(comps[1]))
    ha, hb = comps[:n], (comps[n:] if n > 1 else [comps[1]])
    val = lambda v, c: (ek.detach(v[c]) if ek.__name__.endswith("autodiff") and kind in "fd" else v[c]).numpy()
    s, p = a + b, a * b
    for c in range(n):
        assert bits_equal(val(s, c), (ha[c] + hb[c]).astype(dt)) and bits_equal(val(p, c), (ha[c] * hb[c]).astype(dt))
    lt = a < b
    for c in range(n):
        assert np.array_equal(lt[c].numpy().astype(bool), ha[c] < hb[c])
    if kind in "iu":
        x = a ^ b
        assert np.array_equal(val(x, 0), ha[0] ^ hb[0])
    else:
        q = a / b
        assert bits_equal(val(q, n - 1), (ha[n - 1] / hb[n - 1]).astype(dt))
        d = ek.dot(a, b)
        ref = (ha[0] * hb[0]).astype(dt)
        for c in range(1, n):
            ref = (ha[c].astype(np.float64) * hb[c] + ref).astype(dt) if kind == "d" else np.float32(0) + ref   # order checked below
        assert np.allclose((ek.detach(d) if ek.__name__.endswith("autodiff") else d).numpy(),
                           sum(ha[c].astype(np.float64) * hb[c] for c in range(n)), rtol=1e-5, atol=1e-5)
    # struct gather / scatter through one index array
    idx = rng.integers(0, size, K).astype(np.uint32)
    g = ek.gather(a, ek.UInt32(idx))
    for c in range(n):
        assert bits_equal(val(g, c), ha[c][idx])
    perm = rng.permutation(size).astype(np.uint32)
    tgt = V(A(np.zeros(size, dt)))
    ek.scatter(tgt, a, ek.UInt32(perm))
    for c in range(n):
        expect = np.zeros(size, dt); expect[perm] = ha[c]
        assert bits_equal(val(tgt, c), expect)
    # conversions between the flavours
    other = {"i": "f", "u": "i", "f": "i", "d": "f"}[kind]
    conv = getattr(ek, f"Vector{n}{other}")(a)
    cv = (ek.detach(conv[0]) if ek.__name__.endswith("autodiff") and other in "fd" else conv[0]).numpy()
    assert np.array_equal(cv, ha[0].astype(NP[other]))


@pytest.mark.parametrize("n", [1, 2, 3, 4])
def test_mask_vectors(ek, n):
    rng = np.random.default_rng(n)
    ms = [rng.integers(0, 2, 1000).astype(np.uint8) for _ in range(2 * n)]
    V = getattr(ek, f"Vector{n}m")
    a = V(*[ek.Mask(m) for m in ms[:n]]) if n > 1 else V(ek.Mask(ms[0]))
    b = V(*[ek.Mask(m) for m in ms[n:]]) if n > 1 else V(ek.Mask(ms[1]))
    hb = ms[n:] if n > 1 else [ms[1]]
    assert np.array_equal((a & b)[0].numpy(), ms[0] & hb[0]) and np.array_equal((a | b)[n - 1].numpy(), ms[n - 1] | hb[n - 1])
    assert np.array_equal((~a)[0].numpy(), 1 - ms[0])
    allc = np.ones(1000, np.uint8)
    for c in range(n):
        allc &= ms[c]
    assert np.array_equal(ek.all(a).numpy(), allc)
    assert ek.any_nested(a) == bool(np.any(np.stack(ms[:n])))
    sel = ek.select(ek.Mask(ms[0]), a, b)
    assert np.array_equal(sel[n - 1].numpy(), np.where(ms[0] != 0, ms[n - 1], hb[n - 1]))


@pytest.mark.parametrize("suffix,dt", [("f", np.float32), ("d", np.float64)])
def test_quaternion_matches_reference(ek, suffix, dt):
    z = np.load(os.path.join(GOLDEN, "quaternion.npz"))
    A = ek.Float32 if suffix == "f" else ek.Float64
    Q = getattr(ek, f"Quaternion4{suffix}")
    mk = lambda h: Q(*[A(h[c].astype(dt)) for c in range(4)])
    a, b, t = mk(z["a"]), mk(z["b"]), A(z["t"].astype(dt))
    get = lambda q: np.stack([(ek.detach(q[c]) if ek.__name__.endswith("autodiff") else q[c]).numpy() for c in range(4)])
    out = z["out"]
    na, nb = ek.normalize(a), ek.normalize(b)
    results = {0: a * b, 2: ek.exp(a), 3: ek.log(a), 1: a / b, 4: ek.slerp(na, nb, t), 5: ek.matrix_to_quat(ek.quat_to_matrix3(na)),
               6: ek.sqrt(a), 7: ek.rcp(a)}
    if suffix == "f":
        # the product follows the reference's fma association: bit-exact; the rest contains rcp / rsqrt / division (class C)
        assert bits_equal(get(results[0]), out[0])
    for k, q in results.items():
        tol = 3e-5 if k in (3, 5) else 1e-5
        assert np.allclose(get(q), out[k], rtol=tol, atol=tol), k
    # algebra: q * rcp(q) = 1, rotate() produces unit quaternions, quat_to_matrix is orthonormal
    one = get(a * ek.rcp(a))
    assert np.allclose(one[3], 1, atol=1e-5) and np.allclose(one[:3], 0, atol=1e-5)
    m3 = ek.quat_to_matrix3(na)
    mat = z["mat"]
    for i in range(3):
        for j in range(3):
            e = m3[i, j]
            assert np.allclose((ek.detach(e) if ek.__name__.endswith("autodiff") else e).numpy(), mat[i * 3 + j], atol=2e-6), (i, j)
