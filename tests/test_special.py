"""Special functions (include/enoki/special.h; reference include/enoki/special.h:22-312, tests/special.cpp): erf, erfc,
erfinv, i0e, dawson, erfi, lgamma, tgamma in float32 and float64.

Parity: everything is bit-exact against the reference build except the float32 functions that contain rcp() / rsqrt()
(erf for |x| > 1, erfc, i0e for |x| > 8: rcpps / rsqrtps + Newton on AVX2 versus an exact division -- class C, a few
1e-7 relative).  tests/golden/special.npz was generated from the reference build (tests/golden/make_golden.py)."""
import math
import os

import numpy as np
import pytest

from conftest import bits_equal

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
WIDE = ["erf", "erfc", "i0e", "dawson", "lgamma", "tgamma"]
UNIT = ["erfinv", "erfi"]
CLASS_C_F32 = {"erf": 4e-7, "erfc": 3e-6, "i0e": 1e-6}          # relative bound port / kernel vs AVX2 reference


def same(a, b):
    return bits_equal(a, b) or np.array_equal(np.isnan(a), np.isnan(b)) and bits_equal(np.nan_to_num(a, nan=7.0), np.nan_to_num(b, nan=7.0))


def close(a, b, rel):
    fin = np.isfinite(a) & np.isfinite(b)
    ok = np.abs(a[fin].astype(np.float64) - b[fin]) <= rel * np.abs(a[fin].astype(np.float64)) + 1e-44
    return ok.all() and np.array_equal(a[~fin].astype(np.float64), b[~fin].astype(np.float64), equal_nan=True)


def check(z, tag, fn):
    for op in WIDE + UNIT:
        x = z[f"{tag}_wide"] if op in WIDE else z[f"{tag}_unit"]
        got, want = fn(op, x), z[f"{tag}_{op}"]
        if tag == "f32" and op in CLASS_C_F32:
            assert close(want, got, CLASS_C_F32[op]), (tag, op)
        else:
            assert same(want, got), (tag, op)


def test_oracle_port_matches_golden(oracle):
    z = np.load(os.path.join(GOLDEN, "special.npz"))
    check(z, "f32", oracle.unary)
    check(z, "f64", oracle.unary)


def test_against_libm(oracle):
    """sanity of the approximations themselves (relative accuracy the reference's tests/special.cpp asks for)"""
    x = np.linspace(-3, 3, 2001)
    erf = np.array([math.erf(v) for v in x]); lg = np.array([math.lgamma(v) for v in x + 3.5])
    assert np.allclose(oracle.unary("erf", x), erf, rtol=1e-13, atol=1e-15)
    assert np.allclose(oracle.unary("erfc", x), 1 - erf, rtol=1e-9, atol=1e-15)
    assert np.allclose(oracle.unary("erf", x.astype(np.float32)), erf, rtol=2e-6, atol=1e-7)
    assert np.allclose(oracle.unary("lgamma", (x + 3.5)), lg, rtol=1e-9, atol=1e-9)
    u = np.linspace(-0.99, 0.99, 1001)
    assert np.allclose(oracle.unary("erf", oracle.unary("erfinv", u)), u, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_kernels_bit_exact_vs_oracle_and_golden(capi, oracle):
    from test_kernels_gpu import up
    z = np.load(os.path.join(GOLDEN, "special.npz"))
    fn = lambda op, x: capi.unary(op, up(capi, x)).numpy()
    check(z, "f32", fn)
    check(z, "f64", fn)
    rng = np.random.default_rng(5)
    for dt in (np.float32, np.float64):
        for op in WIDE + UNIT:
            x = (rng.uniform(-12, 12, 100003) if op in WIDE else rng.uniform(-0.9999, 0.9999, 100003)).astype(dt)
            assert same(oracle.unary(op, x), fn(op, x)), (op, dt)


@pytest.mark.gpu
@pytest.mark.parametrize("mod", ["hip", "hip_autodiff"])
def test_python_surface_and_composition(mod, oracle):
    """enoki_amd.hip.*: the fused kernels; enoki_amd.hip_autodiff.*: the generic composition over DiffArray ops -- the same
    operations in the same order, hence the same bits"""
    import importlib
    ek = importlib.import_module(f"enoki_amd.{mod}")
    rng = np.random.default_rng(9)
    for Float, dt in ((ek.Float32, np.float32), (ek.Float64, np.float64)):
        for op in WIDE + UNIT:
            x = (rng.uniform(-10, 10, 4099) if op in WIDE else rng.uniform(-0.999, 0.999, 4099)).astype(dt)
            r = getattr(ek, op)(Float(x))
            r = (ek.detach(r) if mod == "hip_autodiff" else r).numpy()
            assert same(oracle.unary(op, x), r), (mod, op, dt)


@pytest.mark.gpu
def test_gradients_through_the_composition():
    import enoki_amd.hip_autodiff as ek
    x = np.linspace(-2.5, 2.5, 1001).astype(np.float32)
    c = 2.0 / math.sqrt(math.pi)
    cases = {"erf": c * np.exp(-x.astype(np.float64) ** 2), "erfc": -c * np.exp(-x.astype(np.float64) ** 2),
             "erfi": c * np.exp(x.astype(np.float64) ** 2)}
    for op, want in cases.items():
        v = ek.Float32(x); ek.set_requires_gradient(v)
        ek.backward(ek.hsum(getattr(ek, op)(v)))
        assert np.allclose(ek.gradient(v).numpy(), want, rtol=2e-3, atol=2e-4), op
    u = np.linspace(-0.9, 0.9, 501).astype(np.float32)
    v = ek.Float32(u); ek.set_requires_gradient(v)
    y = ek.erfinv(v)
    ek.backward(ek.hsum(y))
    want = math.sqrt(math.pi) / 2 * np.exp(ek.detach(y).numpy().astype(np.float64) ** 2)
    assert np.allclose(ek.gradient(v).numpy(), want, rtol=5e-3)
