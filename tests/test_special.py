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


@pytest.mark.extras
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


@pytest.mark.extras
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


@pytest.mark.extras
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


# ---- elliptic integrals (include/enoki/ellint.h; reference special.h:314-672) ---------------------------------------
ELLINT = ["comp_ellint_1", "comp_ellint_2", "comp_ellint_3", "ellint_1", "ellint_2", "ellint_3", "carlson_rf", "carlson_rd",
          "carlson_rc", "carlson_rj"]


def _ellint_golden(tag):
    z = np.load(os.path.join(GOLDEN, f"ellint_{tag}.npz"))        # from the reference build, tests/golden/make_golden.py
    return z["phi"], z["k"], z["nu"], z["out"]


def test_ellint_host_packets_match_golden():
    """the restated integrals on one-element packets, on the HOST (tests/cpp/ellint_host.cpp): Carlson forms and complete
    integrals are bit-exact in float64; the incomplete ones go through the host's libm sincos here (1 ulp); float32 differs
    by the reference's rcpps-based rcp() (class C)"""
    import ctypes
    lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "libellint_host.so"))
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    for tag, fn in (("f64", lib.ellint_host_f64), ("f32", lib.ellint_host_f32)):
        phi, k, nu, want = _ellint_golden(tag)
        got = np.empty_like(want)
        fn(p(phi), p(k), p(nu), ctypes.c_size_t(phi.size), p(got))
        for i, name in enumerate(ELLINT):
            if tag == "f64" and not name.startswith("ellint_"):
                assert bits_equal(got[i], want[i]), name
            else:
                rel = np.abs(got[i].astype(np.float64) - want[i]) / np.abs(want[i])
                assert rel.max() <= (2e-15 if tag == "f64" else 1e-6), (tag, name, rel.max())
    # and against scipy's independent implementations (k enters squared: m = k^2)
    from scipy import special as sp
    phi, k, nu, want = _ellint_golden("f64")
    assert np.abs(want[3] - sp.ellipkinc(phi, k * k)).max() < 1e-13 and np.abs(want[4] - sp.ellipeinc(phi, k * k)).max() < 1e-13
    assert np.abs(want[0] - sp.ellipk(k * k)).max() < 1e-14 and np.abs(want[1] - sp.ellipe(k * k)).max() < 1e-14
    assert np.abs(want[6] - sp.elliprf(phi * phi, 1.5 - k * k, 1 + np.abs(nu))).max() < 1e-14


@pytest.mark.extras
def test_ellint_device_matches_golden():
    """HIPArray composition (one kernel per operation, device sincos = the reference's algorithm): ALL ten functions are
    bit-exact against the reference build in float64; float32 is class C (rcp)"""
    import enoki_amd.hip as ek
    ek.hip_init(0)
    for tag, T in (("f64", ek.Float64), ("f32", ek.Float32)):
        phi, k, nu, want = _ellint_golden(tag)
        P, K, N = T(phi), T(k), T(nu)
        X, Y, Z, R = P * P, T(1.5) - K * K, T(1.0) + ek.abs(N), T(0.5) + ek.abs(N)
        got = [ek.comp_ellint_1(K), ek.comp_ellint_2(K), ek.comp_ellint_3(K, N), ek.ellint_1(P, K), ek.ellint_2(P, K),
               ek.ellint_3(P, K, N), ek.carlson_rf(X, Y, Z), ek.carlson_rd(X, Y, Z), ek.carlson_rc(X, Y), ek.carlson_rj(X, Y, Z, R)]
        for i, name in enumerate(ELLINT):
            g = got[i].numpy()
            if tag == "f64":
                assert bits_equal(g, want[i]), name
            else:
                rel = np.abs(g.astype(np.float64) - want[i]) / np.abs(want[i])
                assert rel.max() <= 1e-6, (name, rel.max())


@pytest.mark.extras
def test_ellint_gradient():
    """DiffArray differentiates through the duplication rounds: dF/dphi = 1 / sqrt(1 - k^2 sin^2 phi),
    dE/dphi = sqrt(1 - k^2 sin^2 phi)"""
    import enoki_amd.hip_autodiff as ad
    ad.hip_init(0)
    rng = np.random.default_rng(5)
    phi = rng.uniform(-1.4, 1.4, 2000); k = rng.uniform(-0.9, 0.9, 2000)
    for name, deriv in (("ellint_1", lambda d: 1.0 / d), ("ellint_2", lambda d: d)):
        P = ad.Float64(phi); K = ad.Float64(k)
        ad.set_requires_gradient(P)
        y = ad.hsum(getattr(ad, name)(P, K))
        ad.backward(y)
        g = ad.gradient(P).numpy()
        d = np.sqrt(1.0 - (k * np.sin(phi)) ** 2)
        assert np.allclose(g, deriv(d), rtol=2e-5, atol=2e-6), name

