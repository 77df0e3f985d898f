"""The reference's own autodiff test cases (tests/autodiff.cpp: test00 .. test37) re-expressed against
enoki_amd.hip_autodiff: same programs, same closed-form / Mathematica known answers and tolerances the reference
asserts.  Like the reference's `my_backward`, every sweep is preceded by an explicit graph simplification."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ek():
    import enoki_amd.hip_autodiff as m
    m.hip_init(0)
    return m


def bwd(ek, y):
    ek.Float32.simplify_graph()
    ek.backward(y)


def fwd(ek, x):
    ek.Float32.simplify_graph()
    ek.forward(x)


def lin(ek, lo, hi, n, grad=True):
    x = ek.Float32.linspace(lo, hi, n)
    if grad:
        ek.set_requires_gradient(x)
    return x


def g(ek, x):
    return ek.gradient(x).numpy()


def val(ek, x):
    return ek.detach(x).numpy()


def close(a, b, rtol=1e-5, atol=1e-8):          # the reference's allclose defaults
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return bool(np.all(np.abs(a - b) <= atol + rtol * np.abs(b)))


def test00_identity(ek):
    for sweep in (bwd, fwd):
        x = ek.Float32(2.0); ek.set_requires_gradient(x)
        sweep(ek, x)
        assert g(ek, x)[0] == 1.0


def test01_to_04_arithmetic(ek):
    # x = 2, y = 3: z = x (op) y; backward gives (dz/dx, dz/dy); forward(x) then forward(y) give them one at a time
    cases = [(lambda x, y: x + y, (1.0, 1.0)), (lambda x, y: x - y, (1.0, -1.0)), (lambda x, y: x * y, (3.0, 2.0)),
             (lambda x, y: x / y, (1.0 / 3.0, -2.0 / 9.0))]
    for f, (gx, gy) in cases:
        x = ek.Float32(2.0); y = ek.Float32(3.0)
        ek.set_requires_gradient(x); ek.set_requires_gradient(y)
        z = f(x, y)
        bwd(ek, z)
        assert abs(g(ek, x)[0] - gx) < 1e-6 and abs(g(ek, y)[0] - gy) < 1e-6
        x = ek.Float32(2.0); y = ek.Float32(3.0)
        ek.set_requires_gradient(x); ek.set_requires_gradient(y)
        z = f(x, y)
        ek.Float32.simplify_graph()
        ek.forward(x, free_graph=False)
        assert abs(g(ek, z)[0] - gx) < 1e-6
        ek.Float32.simplify_graph()
        ek.forward(y)
        assert abs(g(ek, z)[0] - gy) < 1e-6


def test05_hsum(ek):
    x = lin(ek, 0, 1, 10); y = ek.hsum(x * x); bwd(ek, y)
    assert len(y) == 1 and close(val(ek, y)[0], 95.0 / 27.0) and close(g(ek, x), 2 * val(ek, x))
    x = lin(ek, 0, 1, 10); y = ek.hsum(x * x); fwd(ek, x)
    assert close(g(ek, y), 10)
    x = lin(ek, 0, 1, 11); z = ek.hsum(ek.hsum(x) * x); bwd(ek, z)
    assert np.all(g(ek, x) == 11.0)
    x = lin(ek, 0, 1, 10); y = ek.hsum(ek.hsum(x) * x); fwd(ek, x)
    assert close(g(ek, y), 100)
    x = lin(ek, 0, 1, 11); z = ek.hsum(ek.hsum(x * x) * x * x); bwd(ek, z)
    assert close(g(ek, x), [0, 1.54, 3.08, 4.62, 6.16, 7.7, 9.24, 10.78, 12.32, 13.86, 15.4])
    x = lin(ek, 0, 1, 10); y = ek.hsum(ek.hsum(x * x) * ek.hsum(x * x)); fwd(ek, x)
    assert close(g(ek, y), 1900.0 / 27.0)


def test06_hprod(ek):
    x = lin(ek, 1, 2, 10); y = ek.hprod(x); bwd(ek, y)
    xv = val(ek, x).astype(np.float64)
    assert len(y) == 1 and close(val(ek, y)[0], 45.5402) and close(g(ek, x), np.prod(xv) / xv)


UNARY = [  # name, domain, derivative (the expressions asserted by test07 .. test27)
    ("sqrt", (1, 2), lambda x: 0.5 / np.sqrt(x)), ("rsqrt", (1, 2), lambda x: -0.5 * x ** -1.5),
    ("exp", (0, 1), np.exp), ("log", (0.01, 1), lambda x: 1 / x), ("sin", (0, 1), np.cos),
    ("cos", (0.01, 1), lambda x: -np.sin(x)), ("tan", (0, 1), lambda x: 1 / np.cos(x) ** 2),
    ("csc", (1, 2), lambda x: -1 / (np.sin(x) * np.tan(x))), ("sec", (1, 2), lambda x: np.tan(x) / np.cos(x)),
    ("asin", (-0.8, 0.8), lambda x: 1 / np.sqrt(1 - x * x)), ("acos", (-0.8, 0.8), lambda x: -1 / np.sqrt(1 - x * x)),
    ("atan", (-0.8, 0.8), lambda x: 1 / (1 + x * x)), ("sinh", (-1, 1), np.cosh), ("cosh", (-1, 1), np.sinh),
    ("tanh", (-1, 1), lambda x: 1 / np.cosh(x) ** 2), ("csch", (1, 2), lambda x: -1 / (np.sinh(x) * np.tanh(x))),
    ("sech", (-1, 1), lambda x: -np.tanh(x) / np.cosh(x)), ("coth", (1, 2), lambda x: 1 - 1 / np.tanh(x) ** 2),
    ("acosh", (1.01, 2), lambda x: 1 / np.sqrt(x * x - 1)), ("atanh", (-0.99, 0.99), lambda x: 1 / (1 - x * x)),
]


@pytest.mark.parametrize("name,domain,deriv", UNARY, ids=[u[0] for u in UNARY])
def test07_to_27_unary(ek, name, domain, deriv):
    x = lin(ek, domain[0], domain[1], 10)
    f = getattr(ek, name) if hasattr(ek, name) else (lambda v: ek.rcp(ek.tanh(v)))
    y = f(x)
    bwd(ek, y)                                                  # vector output: seed of ones
    xv = val(ek, x).astype(np.float64)
    assert close(g(ek, x), deriv(xv), rtol=1e-5, atol=1e-6), name


def test28_linear_to_srgb(ek):
    x = lin(ek, 0, 1, 10)
    # color.h linear_to_srgb: x <= 0.0031308 ? 12.92 x : 1.055 x^(1/2.4) - 0.055
    y = ek.select(x <= ek.Float32(0.0031308), x * ek.Float32(12.92),
                  ek.fmadd(ek.Float32(1.055), ek.pow(x, ek.Float32(1.0 / 2.4)), ek.Float32(-0.055)))
    bwd(ek, y)
    ref = [12.92, 1.58374, 1.05702, 0.834376, 0.705474, 0.61937, 0.556879, 0.50899, 0.470847, 0.439583]   # Mathematica
    assert np.abs(g(ek, x) - np.array(ref)).max() < 1e-5


def test29_scatter_add(ek):
    idx1 = ek.UInt32.arange(5); idx2 = ek.UInt32.arange(4) + ek.UInt32(3)
    x = lin(ek, 0, 1, 5); y = lin(ek, 1, 2, 4)
    buf = ek.Float32.zero(10)
    ek.scatter_add(buf, x, idx1); ek.scatter_add(buf, y, idx2)
    assert close(val(ek, buf), [0, 0.25, 0.5, 1.75, 2.3333, 1.6667, 2.0, 0, 0, 0], 1e-4, 1e-4)
    s = ek.hsum(buf * buf)
    bwd(ek, s)
    assert close(g(ek, y), [3.5, 4.6667, 3.3333, 4.0], 1e-4, 1e-4)
    assert close(g(ek, x), [0, 0.5, 1.0, 3.5, 4.6667], 1e-4, 1e-4)


def test30_scatter(ek):
    idx1 = ek.UInt32.arange(5); idx2 = ek.UInt32.arange(4) + ek.UInt32(3)
    x = lin(ek, 0, 1, 5); y = lin(ek, 1, 2, 4)
    buf = ek.Float32.zero(10)
    ek.scatter(buf, x, idx1); ek.scatter(buf, y, idx2)
    assert close(val(ek, buf), [0, 0.25, 0.5, 1.0, 1.3333, 1.6667, 2.0, 0, 0, 0], 1e-4, 1e-4)
    s = ek.hsum(buf * buf)
    bwd(ek, s)
    assert close(g(ek, y), [2.0, 2.6667, 3.3333, 4.0], 1e-4, 1e-4)
    assert close(g(ek, x), [0, 0.5, 1.0, 0, 0], 1e-4, 1e-4)


def test33_bcast(ek):
    x = ek.Float32(5.0); y = ek.Float32.arange(10)
    ek.set_requires_gradient(x); ek.set_requires_gradient(y)
    ek.set_label(x, "x"); ek.set_label(y, "y")
    t = ek.sin(x) * ek.cos(y)
    z = ek.hsum(t * t)
    bwd(ek, z)
    assert close(g(ek, x), -2.8803, 1e-4, 1e-4)
    assert close(g(ek, y), [-0.0, -0.8361, 0.6959, 0.2569, -0.9098, 0.5002, 0.4934, -0.9109, 0.2647, 0.6906], 1e-4, 1e-4)


def test34_gradient_descent(ek):
    x = ek.Float32.zero(10)
    loss_f = 0.0
    for _ in range(10):
        ek.set_requires_gradient(x)
        d = x - ek.Float32.linspace(0, 1, 10)
        loss = ek.sqrt(ek.hsum(d * d))                         # norm()
        bwd(ek, loss)
        x = ek.Float32(ek.detach(x) - ek.gradient(x) * type(ek.gradient(x))(2e-1))
        loss_f = float(val(ek, loss)[0])
    assert loss_f < 1e-1


def test36_gather(ek):
    x = lin(ek, -1, 1, 10)
    y = ek.gather(x * x, ek.UInt32(np.array([1, 2, 3], np.uint32)))
    z = ek.hsum(y)
    bwd(ek, z)
    assert close(g(ek, x), [0, -1.55556, -1.11111, -0.666667, 0, 0, 0, 0, 0, 0], 1e-4, 1e-4)
    x = lin(ek, -1, 1, 10)
    y = ek.gather(x * x, ek.UInt32(np.array([1, 2, 3], np.uint32)))
    fwd(ek, x)
    assert close(g(ek, y), [-1.55556, -1.11111, -0.666667], 1e-4, 1e-4)


def test37_scatter_fwd(ek):
    x = lin(ek, -1, 1, 5)
    y = ek.Float32.zero(10)
    ek.scatter(y, x * x, ek.UInt32.arange(5) + ek.UInt32(2))
    fwd(ek, x)
    assert close(g(ek, y), [0, 0, -2, -1, 0, 1, 2, 0, 0, 0], 1e-4, 1e-4)
