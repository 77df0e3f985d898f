"""Pin the CPU oracle (oracle/enoki_oracle.c, "port") two ways:
  1. live, bit-for-bit against the UNMODIFIED reference build oracle/_ref (only where it exists);
  2. against the committed golden vectors in tests/golden/ (generated from that reference build by
     tests/golden/make_golden.py) -- this is what keeps the oracle pinned on machines without /root/reference.
Known, documented deviations of the AVX2 reference path that the oracle does NOT copy:
  * rcp / rsqrt: rcpps / rsqrtps + one Newton step (array_avx.h:324-395), ISA specific -> class C, see below;
  * u32 -> f32 conversion double-rounds above 2^31 (array_avx.h:56-66); scalar/AVX-512/CUDA paths round correctly;
  * integer division, f32 <-> 64-bit integer casts and 64-bit arithmetic shifts >= 64 of DynamicArray<Packet<T,8>>
    return indeterminate values in this reference build (padding lanes) -> defined by C semantics instead.
"""
import os

import numpy as np
import pytest

import oracle_lib as ol
from conftest import bits_equal, f32_inputs, f64_inputs, ulp_diff, uniform_pm1, hash_u32

HAVE_REF = os.path.exists(os.path.join(ol.ORACLE_DIR, "_ref", "libenoki_ref.so"))
needs_ref = pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref was not built (needs /root/reference)")
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
P = ol.port()


@needs_ref
@pytest.mark.parametrize("scale", [1.0, 30.0, 3000.0])
def test_f32_vertical_ops_bit_exact(scale):
    R = ol.ref()
    a = f32_inputs(200003, 1, scale); b = f32_inputs(200003, 2, scale)[::-1].copy(); c = f32_inputs(200003, 3)
    for op in ["neg", "abs", "sqrt", "floor", "ceil", "round", "trunc", "sin", "cos", "exp", "log", "sign"]:
        assert bits_equal(P.unary(op, a), R.unary(op, a)), op
    ps, pc = P.sincos(a); rs, rc = R.sincos(a)
    assert bits_equal(ps, rs) and bits_equal(pc, rc)
    for op in ["add", "sub", "mul", "div", "min", "max", "safe_mul"]:
        assert bits_equal(P.binary(op, a, b), R.binary(op, a, b)), op
    for op in ["fmadd", "fmsub", "fnmadd", "fnmsub", "safe_fmadd"]:
        assert bits_equal(P.ternary(op, a, b, c), R.ternary(op, a, b, c)), op
    for op in ["eq", "neq", "lt", "le", "gt", "ge"]:
        assert np.array_equal(P.compare(op, a, b), R.compare(op, a, b)), op


@needs_ref
@pytest.mark.parametrize("scale", [1.0, 30.0, 3000.0, 1e6])
def test_f64_transcendentals_bit_exact(scale):
    """double branches of sin/cos/sincos/exp/log.  sin/cos only for |x| <= 3e9: beyond |x|*4/pi = 2^32 this reference
    build returns indeterminate values (its int64 packets convert through 32-bit lanes without AVX512DQ)."""
    R = ol.ref()
    a = f64_inputs(200003, 31, scale, limit=3e9)
    for op in ["sin", "cos", "exp", "log"]:
        assert bits_equal(P.unary(op, a), R.unary(op, a)), op
    ps, pc = P.sincos(a); rs, rc = R.sincos(a)
    assert bits_equal(ps, rs) and bits_equal(pc, rc)
    b = f64_inputs(200003, 32, 1e300)                  # exp / log have no such restriction
    for op in ["exp", "log"]:
        assert bits_equal(P.unary(op, b), R.unary(op, b)), op
    x = np.abs(f64_inputs(100000, 33, 10.0, specials=False)) + 1e-3
    assert np.abs(P.unary("log", x) - np.log(x)).max() < 2e-15 and np.abs(P.unary("sin", x) - np.sin(x)).max() < 3e-16


@needs_ref
@pytest.mark.parametrize("scale", [0.3, 1.0, 30.0, 3000.0])
def test_f64_second_wave_bit_exact(scale):
    """double branches of tan .. cbrt, atan2, pow, fmod, ldexp: all bit-exact -- AVX2 has no rcppd, so even the
    functions that call rcp() divide exactly in the reference build"""
    R = ol.ref()
    a = f64_inputs(100003, 41, scale, limit=3e9); b = f64_inputs(100003, 42, scale, limit=3e9)[::-1].copy()
    for op in ["tan", "cot", "asin", "acos", "atan", "sinh", "cosh", "tanh", "asinh", "acosh", "atanh", "cbrt"]:
        assert bits_equal(P.unary(op, a), R.unary(op, a)), op
    for op in ["atan2", "pow", "fmod"]:
        assert bits_equal(P.binary(op, a, b), R.binary(op, a, b)), op
    e = np.clip(np.trunc(b), -500, 500)
    assert bits_equal(P.binary("ldexp", a, e), R.binary("ldexp", a, e))


CLASS_A2 = ["asin", "acos", "atan", "asinh", "acosh", "atanh", "cbrt"]   # no rcp() inside: bit-exact
CLASS_C2 = {"tan": 8, "cot": 8, "sinh": 8, "cosh": 8, "tanh": 16}        # rcp() inside: ulp bound port vs reference


@needs_ref
@pytest.mark.parametrize("scale", [0.3, 1.0, 30.0, 3000.0])
def test_second_wave_bit_exact(scale):
    """array_math.h second wave.  Functions that do not call rcp() match the reference bit for bit."""
    R = ol.ref()
    a = f32_inputs(200003, 21, scale); b = f32_inputs(200003, 22, scale)[::-1].copy()
    for op in CLASS_A2:
        assert bits_equal(P.unary(op, a), R.unary(op, a)), op
    for op in ["atan2", "pow", "fmod"]:
        assert bits_equal(P.binary(op, a, b), R.binary(op, a, b)), op
    e = np.clip(np.trunc(b), -100, 100).astype(np.float32)
    assert bits_equal(P.binary("ldexp", a, e), R.binary("ldexp", a, e))


@needs_ref
def test_second_wave_class_c():
    """tan/cot/sinh/cosh/tanh contain rcp(): the AVX2 reference uses rcpps + one Newton step (array_avx.h:324-357,
    ISA specific), the oracle an exact division -> a few ulp apart away from overflow / denormal results."""
    R = ol.ref()
    rng = np.random.default_rng(5)
    for op, bound in CLASS_C2.items():
        a = rng.uniform(-9, 9, 200000).astype(np.float32)
        p, r = P.unary(op, a), R.unary(op, a)
        ok = np.isfinite(r) & (np.abs(r) > 1e-30) & (np.abs(r) < 1e30)
        assert ok.mean() > 0.99 and ulp_diff(p[ok], r[ok]).max() <= bound, op
        truth = {"tan": np.tan, "cot": lambda x: 1 / np.tan(x), "sinh": np.sinh, "cosh": np.cosh, "tanh": np.tanh}[op](
            a.astype(np.float64))
        rel = np.abs(p.astype(np.float64) - truth) / np.maximum(np.abs(truth), 1e-30)
        assert np.percentile(rel, 99.9) < 2e-6, op          # the algorithm's own accuracy (array_math.h:376-386)


@needs_ref
def test_rcp_rsqrt_class_c():
    """both the IEEE oracle and the AVX2 reference stay within the reference's own test bounds vs float64"""
    R = ol.ref()
    a = np.exp(np.random.default_rng(0).uniform(-80, 80, 100000)).astype(np.float32)
    for op, bound, truth in (("rcp", 2, 1.0 / a.astype(np.float64)), ("rsqrt", 3, 1.0 / np.sqrt(a.astype(np.float64)))):
        t = truth.astype(np.float32)
        assert ulp_diff(P.unary(op, a), t).max() <= 1
        assert ulp_diff(R.unary(op, a), t).max() <= bound + 1


@needs_ref
@pytest.mark.parametrize("dt", [np.int32, np.uint32, np.int64, np.uint64])
def test_integer_ops_bit_exact(dt):
    R = ol.ref()
    rng = np.random.default_rng(2); n = 100000; info = np.iinfo(dt)
    a = rng.integers(info.min, info.max, n, dtype=dt, endpoint=True)
    b = rng.integers(info.min, info.max, n, dtype=dt, endpoint=True)
    for op in ["neg", "not", "abs", "popcnt", "lzcnt", "tzcnt"]:
        assert np.array_equal(P.unary(op, a), R.unary(op, a)), op
    for op in ["add", "sub", "mul", "min", "max", "mulhi", "and", "or", "xor"]:
        assert np.array_equal(P.binary(op, a, b), R.binary(op, a, b)), op
    bits = 8 * np.dtype(dt).itemsize
    sh = rng.integers(0, bits, n).astype(dt)
    for op in ["sl", "sr"]:
        assert np.array_equal(P.binary(op, a, sh), R.binary(op, a, sh)), op
    if bits == 32:
        sh2 = rng.integers(0, 40, n).astype(dt)
        for op in ["sl", "sr"]:
            assert np.array_equal(P.binary(op, a, sh2), R.binary(op, a, sh2)), op
        for op in ["fmadd", "fmsub", "fnmadd", "fnmsub"]:
            assert np.array_equal(P.ternary(op, a, b, a[::-1].copy()), R.ternary(op, a, b, a[::-1].copy())), op
    for op in ["eq", "neq", "lt", "le", "gt", "ge"]:
        assert np.array_equal(P.compare(op, a, b), R.compare(op, a, b)), op
    m = rng.integers(0, 2, n).astype(np.uint8)
    assert np.array_equal(P.select(m, a, b), R.select(m, a, b))
    for op in ["hsum", "hprod", "hmin", "hmax"]:
        for nn in [0, 1, 2, 7, 8, 9, 1000]:
            assert P.reduce(op, a[:nn]) == R.reduce(op, a[:nn]), (op, nn)


@needs_ref
def test_casts():
    R = ol.ref()
    rng = np.random.default_rng(3); n = 100000
    f = (rng.standard_normal(n) * 1e3).astype(np.float32); f[:6] = [0.5, -0.5, 1.5, -1.5, 2.5, -2.5]
    assert np.array_equal(P.cast(f, np.int32), R.cast(f, np.int32))
    assert np.array_equal(P.cast(np.abs(f), np.uint32), R.cast(np.abs(f), np.uint32))
    assert bits_equal(P.cast(f, np.float64), R.cast(f, np.float64))
    big = np.array([3e9, -3e9, np.nan, np.inf, -np.inf, 2147483520.0, 2147483648.0, -2147483648.0], np.float32)
    assert np.array_equal(P.cast(big, np.int32), R.cast(big, np.int32))
    i = rng.integers(-2**31, 2**31 - 1, n, dtype=np.int32)
    assert bits_equal(P.cast(i, np.float32), R.cast(i, np.float32))
    u = rng.integers(0, 2**32 - 1, n, dtype=np.uint32)
    small = u >> np.uint32(1)
    assert bits_equal(P.cast(small, np.float32), R.cast(small, np.float32))
    assert ulp_diff(P.cast(u, np.float32), R.cast(u, np.float32)).max() <= 1      # AVX2 double rounding above 2^31


@needs_ref
@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 63, 64, 65, 1000, 100003])
def test_memory_and_horizontal_ops(n):
    R = ol.ref()
    rng = np.random.default_rng(n); K = 257
    src = rng.standard_normal(K).astype(np.float32)
    idx = rng.integers(0, K, n).astype(np.uint32); m = (rng.integers(0, 4, n) != 0).astype(np.uint8)
    val = rng.standard_normal(n).astype(np.float32)
    for it in (np.uint32, np.int32):
        assert bits_equal(P.gather(src, idx.astype(it), m), R.gather(src, idx.astype(it), m))
        assert bits_equal(P.scatter(src, val, idx.astype(it), m), R.scatter(src, val, idx.astype(it), m))
        assert bits_equal(P.scatter(src, val, idx.astype(it), m, add=True), R.scatter(src, val, idx.astype(it), m, add=True))
    for op in ["hsum", "hprod", "hmin", "hmax"]:
        aa = val if op != "hprod" else (1 + 0.01 * val).astype(np.float32)
        assert bits_equal(np.float32(P.reduce(op, aa)), np.float32(R.reduce(op, aa))), (op, n)   # same packet order
    for op in ["all", "any", "count"]:
        for mm in (m, np.ones(n, np.uint8), np.zeros(n, np.uint8)):
            assert P.mask_reduce(op, mm) == R.mask_reduce(op, mm)
    if n > 1:
        assert bits_equal(P.linspace(-1.2, 1.2, n), R.linspace(-1.2, 1.2, n))
    assert bits_equal(P.psum(val), R.psum(val))


@needs_ref
def test_empty_reductions():
    R = ol.ref()
    e = np.zeros(0, np.float32)
    for op in ["hsum", "hprod", "hmin", "hmax"]:
        assert bits_equal(np.float32(P.reduce(op, e)), np.float32(R.reduce(op, e)))
    for op in ["all", "any", "count"]:
        assert P.mask_reduce(op, np.zeros(0, np.uint8)) == R.mask_reduce(op, np.zeros(0, np.uint8))


@needs_ref
@pytest.mark.parametrize("n", [1000, 100003, 1 << 20])
def test_baseline_configs_bit_exact(n):
    R = ol.ref()
    a, x, b = uniform_pm1(n, 1), uniform_pm1(n, 2), uniform_pm1(n, 3)
    assert P.cfg1(a, x, b)[0] == R.cfg1(a, x, b)[0]
    assert P.cfg2(a, x, b)[0] == R.cfg2(a, x, b)[0]
    py, pga, pgb, _ = P.cfg3a(a, x, b); ry, rga, rgb, _ = R.cfg3a(a, x, b)
    assert py == ry and bits_equal(pga, rga) and bits_equal(pgb, rgb)
    K = 4096
    A, B = uniform_pm1(K, 6), uniform_pm1(K, 7)
    idx = (hash_u32(np.arange(n, dtype=np.uint64), 4) % np.uint32(K)).astype(np.uint32)
    py, pga, pgb, _ = P.cfg3b(A, B, x, idx); ry, rga, rgb, _ = R.cfg3b(A, B, x, idx)
    assert py == ry and bits_equal(pga, rga) and bits_equal(pgb, rgb)


# ---- golden fixtures (always available) --------------------------------------------------------------
def test_golden_elementwise():
    z = np.load(os.path.join(GOLDEN, "elementwise_f32.npz"))
    a, b, c = z["in_a"], z["in_b"], z["in_c"]
    for key in z.files:
        kind, _, op = key.partition("_")
        if kind == "unary":
            assert bits_equal(P.unary(op, a), z[key]), key
        elif kind == "binary":
            assert bits_equal(P.binary(op, a, b), z[key]), key
        elif kind == "ternary":
            assert bits_equal(P.ternary(op, a, b, c), z[key]), key
        elif kind == "compare":
            assert np.array_equal(P.compare(op, a, b), z[key]), key
    s, co = P.sincos(a)
    assert bits_equal(s, z["sincos_s"]) and bits_equal(co, z["sincos_c"])


def test_golden_second_wave():
    z = np.load(os.path.join(GOLDEN, "elementwise2_f32.npz"))
    a, b = z["in_a"], z["in_b"]
    for op in CLASS_A2:
        assert bits_equal(P.unary(op, a), z[f"unary_{op}"]), op
    for op in ["asin", "acos", "atanh"]:
        assert bits_equal(P.unary(op, z["in_unit"]), z[f"unit_{op}"]), op
    for op in ["atan2", "pow", "fmod"]:
        assert bits_equal(P.binary(op, a, b), z[f"binary_{op}"]), op
    assert bits_equal(P.binary("ldexp", z["in_c"], np.clip(z["in_e"], -100, 100)), z["ldexp"])
    for op, bound in CLASS_C2.items():
        p, r = P.unary(op, a), z[f"unary_{op}"]
        ok = np.isfinite(r) & np.isfinite(p) & (np.abs(r) > 1e-30) & (np.abs(r) < 1e30) & (np.abs(a) < 50)
        assert ulp_diff(p[ok], r[ok]).max() <= bound, op


def test_golden_f64():
    z = np.load(os.path.join(GOLDEN, "elementwise_f64.npz"))
    d = z["in_d"]
    for op in ["sin", "cos", "exp", "log"]:
        assert bits_equal(P.unary(op, d), z[op]), op
    assert bits_equal(P.unary("log", z["in_pos"]), z["log_pos"])
    s_, c_ = P.sincos(d)
    assert bits_equal(s_, z["sincos_s"]) and bits_equal(c_, z["sincos_c"])
    for op in ["tan", "cot", "atan", "sinh", "cosh", "tanh", "asinh", "cbrt"]:
        assert bits_equal(P.unary(op, d), z[f"sw_{op}"]), op
    for op in ["asin", "acos", "atanh"]:
        assert bits_equal(P.unary(op, z["in_unit"]), z[f"sw_{op}"]), op
    assert bits_equal(P.unary("acosh", z["in_pos"]), z["sw_acosh"])
    for op in ["atan2", "pow", "fmod"]:
        assert bits_equal(P.binary(op, d, z["in_d2"]), z[f"sw_{op}"]), op


def test_golden_integer():
    z = np.load(os.path.join(GOLDEN, "integer.npz"))
    for t in ("int32", "uint32"):
        x, y, sh = z[f"{t}_x"], z[f"{t}_y"], z[f"{t}_sh"]
        for key in z.files:
            if not key.startswith(t + "_") or key.count("_") < 2:
                continue
            _, kind, op = key.split("_", 2)
            if kind == "unary":
                assert np.array_equal(P.unary(op, x), z[key]), key
            elif kind == "binary":
                assert np.array_equal(P.binary(op, x, sh if op in ("sl", "sr") else y), z[key]), key


def test_golden_memory_reduce_configs():
    z = np.load(os.path.join(GOLDEN, "memory_reduce.npz"))
    assert bits_equal(P.gather(z["src"], z["idx"], z["mask"]), z["gather"])
    assert bits_equal(P.scatter(z["src"], z["val"], z["idx"], z["mask"], add=True), z["scatter_add"])
    for nn in (0, 1, 7, 8, 9, 1000):
        for op in ("hsum", "hprod", "hmin", "hmax"):
            v = z["val"][:nn]
            got = P.reduce(op, (1 + 0.01 * v).astype(np.float32) if op == "hprod" else v)
            assert bits_equal(np.float32(got), z[f"{op}_{nn}"][0]), (op, nn)
    c = np.load(os.path.join(GOLDEN, "configs.npz"))
    for nn in (1000, 65536):
        A, X, B = uniform_pm1(nn, 1), uniform_pm1(nn, 2), uniform_pm1(nn, 3)
        assert np.float32(P.cfg1(A, X, B)[0]) == c[f"cfg1_{nn}"][0]
        assert np.float32(P.cfg2(A, X, B)[0]) == c[f"cfg2_{nn}"][0]
        y, ga, gb, _ = P.cfg3a(A, X, B)
        assert np.float32(y) == c[f"cfg3a_{nn}_y"][0] and bits_equal(ga, c[f"cfg3a_{nn}_ga"]) and bits_equal(gb, c[f"cfg3a_{nn}_gb"])
        Kc = 1024
        TA, TB = uniform_pm1(Kc, 6), uniform_pm1(Kc, 7)
        I = (hash_u32(np.arange(nn, dtype=np.uint64), 4) % np.uint32(Kc)).astype(np.uint32)
        y, gA, gB, _ = P.cfg3b(TA, TB, X, I)
        assert np.float32(y) == c[f"cfg3b_{nn}_y"][0] and bits_equal(gA, c[f"cfg3b_{nn}_gA"]) and bits_equal(gB, c[f"cfg3b_{nn}_gB"])


def test_class_c_pinned_to_the_scalar_row_and_between_the_rows():
    """rcp / rsqrt / division of the oracle: bit-exact against the reference's SCALAR row (oracle/Makefile refscalar: rcp =
    1 / a); tan ... i0e: no further from either row of the reference than the rows are from each other (conftest.py,
    CLASS_C_BAND).  From the committed fixture, and live against oracle/_ref/libenoki_refscalar.so where it exists."""
    from conftest import CLASS_C_BAND, CLASS_C_EXACT, class_c_arg, class_c_check, class_c_fixture
    import oracle_lib as ol
    z = class_c_fixture()
    P = ol.port()
    for op in CLASS_C_EXACT + list(CLASS_C_BAND):
        class_c_check(op, P.unary(op, class_c_arg(op, z)), z)
    assert bits_equal(P.binary("div", z["x"], z["y"]), z["scalar_div"]) and bits_equal(z["scalar_div"], z["avx2_div"])
    try:
        S = ol.ref_scalar()
    except (FileNotFoundError, OSError):
        return
    for op in CLASS_C_EXACT + list(CLASS_C_BAND):               # the fixture is what the live build produces
        assert bits_equal(S.unary(op, class_c_arg(op, z)), z[f"scalar_{op}"]), op
    a = f32_inputs(100003, seed=77, scale=50.0)
    assert bits_equal(P.unary("rcp", a), S.unary("rcp", a))
    assert bits_equal(P.unary("rsqrt", np.abs(a)), S.unary("rsqrt", np.abs(a)))
