"""Complex<Value> over device arrays (include/enoki/complex.h; reference include/enoki/complex.h:60-200).
tests/golden/complex.npz holds a*b, a/b, exp, log, sqrt, pow, sin, cos, rcp, abs, arg of the reference build
(oracle/ref_driver.cpp:ref_complex).  Products, exp, log (its real part), sin, cos follow the reference's operation
order and must agree bit for bit; everything that goes through rcp() (division, rcp, the sinh/cosh inside sin/cos)
or through atan2's rcp-free-but-division path is compared accordingly (class C: a few ulp)."""
import os

import numpy as np
import pytest

from conftest import bits_equal

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MUL, DIV, EXP, LOG, SQRT, POW, SIN, COS, RCP, ABSARG = range(10)


def test_restated_product_order(oracle):
    """CPU: re = fmsub(re0, re1, im0 * im1), im = fmadd(re0, im1, im0 * re1) -- the reference's fmaddsub formulation"""
    z = np.load(os.path.join(GOLDEN, "complex.npz"))
    a, b, out = z["a"], z["b"], z["out"]
    assert bits_equal(oracle.ternary("fmsub", a[0], b[0], a[1] * b[1]), out[MUL][0])
    assert bits_equal(oracle.ternary("fmadd", a[0], b[1], a[1] * b[0]), out[MUL][1])
    e = oracle.unary("exp", a[0])
    s, c = oracle.sincos(a[1])
    assert bits_equal(e * c, out[EXP][0]) and bits_equal(e * s, out[EXP][1])
    sq = oracle.ternary("fmadd", a[1], a[1], a[0] * a[0])
    assert bits_equal(np.float32(0.5) * oracle.unary("log", sq), out[LOG][0])
    assert bits_equal(oracle.binary("atan2", a[1], a[0]), out[LOG][1])
    assert bits_equal(np.sqrt(sq), out[ABSARG][0])


@pytest.mark.extras
@pytest.mark.parametrize("mod", ["hip", "hip_autodiff"])
def test_complex_ops_match_reference(mod):
    import importlib
    ek = importlib.import_module(f"enoki_amd.{mod}")
    z = np.load(os.path.join(GOLDEN, "complex.npz"))
    out = z["out"]
    a = ek.Complex2f(ek.Float32(z["a"][0]), ek.Float32(z["a"][1]))
    b = ek.Complex2f(ek.Float32(z["b"][0]), ek.Float32(z["b"][1]))
    num = lambda x: (ek.detach(x) if mod == "hip_autodiff" else x).numpy()
    pair = lambda c: (num(c.real), num(c.imag))
    exact = {MUL: a * b, EXP: ek.exp(a), LOG: ek.log(a)}
    for k, v in exact.items():
        re, im = pair(v)
        assert bits_equal(re, out[k][0]) and bits_equal(im, out[k][1]), k
    assert bits_equal(num(ek.abs(a)), out[ABSARG][0]) and bits_equal(num(ek.arg(a)), out[ABSARG][1])
    close = {DIV: a / b, RCP: ek.rcp(a), SQRT: ek.sqrt(a), SIN: ek.sin(a), COS: ek.cos(a), POW: ek.pow(a, b)}
    for k, v in close.items():
        re, im = pair(v)
        tol = 2e-5 if k == POW else 4e-6
        assert np.allclose(re, out[k][0], rtol=tol, atol=tol) and np.allclose(im, out[k][1], rtol=tol, atol=tol), k
    # algebra: (a / b) * b = a, sqrt(a)^2 = a, exp(log(a)) = a
    for got in (pair((a / b) * b), pair(ek.sqrt(a) * ek.sqrt(a)), pair(ek.exp(ek.log(a)))):
        assert np.allclose(got[0], z["a"][0], rtol=1e-5, atol=1e-5) and np.allclose(got[1], z["a"][1], rtol=1e-5, atol=1e-5)
    assert bits_equal(num(ek.conj(a).imag), -z["a"][1]) and bits_equal(num((a * ek.Float32(2.0)).real), z["a"][0] * np.float32(2))


@pytest.mark.extras
def test_complex_gradient():
    """d/d re |z|^2 = 2 re through the complex product z * conj(z)"""
    import enoki_amd.hip_autodiff as ek
    z = np.load(os.path.join(GOLDEN, "complex.npz"))
    re = ek.Float32(z["a"][0]); im = ek.Float32(z["a"][1])
    ek.set_requires_gradient(re); ek.set_requires_gradient(im)
    c = ek.Complex2f(re, im)
    p = c * ek.conj(c)
    ek.backward(ek.hsum(p.real))
    assert np.allclose(ek.gradient(re).numpy(), 2 * z["a"][0], rtol=1e-6)
    assert np.allclose(ek.gradient(im).numpy(), 2 * z["a"][1], rtol=1e-6)


MORE = ["sinh", "cosh", "tanh", "asin", "acos", "atan", "asinh", "acosh", "atanh"]


def test_complex_more_golden_is_sane():
    """CPU: the reference's outputs for the hyperbolic / inverse functions against numpy's complex functions (acosh only where
    the two branch-cut conventions agree: the reference evaluates log(z + sqrt(z*z - 1)))"""
    z = np.load(os.path.join(GOLDEN, "complex_more.npz"))
    a = z["a"][0].astype(np.float64) + 1j * z["a"][1]
    fns = [np.sinh, np.cosh, np.tanh, np.arcsin, np.arccos, np.arctan, np.arcsinh, np.arccosh, np.arctanh]
    for k, f in enumerate(fns):
        g = z["out"][k][0].astype(np.float64) + 1j * z["out"][k][1]
        sel = (a.real > 0) if MORE[k] == "acosh" else np.ones(a.shape, bool)
        assert np.abs(g[sel] - f(a[sel])).max() < 5e-6, MORE[k]


@pytest.mark.extras
def test_complex_more_matches_reference():
    """sinh ... atanh of Complex2f against the reference build (tests/golden/complex_more.npz): the same compositions of
    log / sqrt / sincos / sincosh; class C where rcp() or a division enters"""
    import enoki_amd.hip as ek
    z = np.load(os.path.join(GOLDEN, "complex_more.npz"))
    a = ek.Complex2f(ek.Float32(z["a"][0]), ek.Float32(z["a"][1]))
    for k, name in enumerate(MORE):
        r = getattr(ek, name)(a)
        re, im = r.real.numpy(), r.imag.numpy()
        assert np.allclose(re, z["out"][k][0], rtol=2e-5, atol=2e-5) and np.allclose(im, z["out"][k][1], rtol=2e-5, atol=2e-5), name

