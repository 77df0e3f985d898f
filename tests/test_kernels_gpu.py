"""Parity of the HIP kernels (called THROUGH the C ABI, enoki_amd/capi.py -> libenoki-hip.so) against
the CPU oracle (oracle/enoki_oracle.c, itself pinned bit-exactly to the reference build, see
tests/test_oracle_vs_ref.py).

Parity classes:
  A  bit-exact        integer/mask/index ops, IEEE arithmetic, rounding, casts, gather/scatter,
                      sin/cos/exp/log (restated CEPHES with explicit fma)
  C  vs float64 truth rcp <= 2 ulp, rsqrt <= 3 ulp (the reference's own test bounds, tests/float.cpp:129-165;
                      the AVX2 reference uses rcpps/rsqrtps + one Newton step, which is ISA specific)
  D  order dependent  hsum/hprod/fp scatter_add: |gpu - f64 truth| <= gamma_n * sum|x_i|
"""
import numpy as np
import pytest

from conftest import bits_equal, f32_inputs, f64_inputs, ulp_diff

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 1000, 4099, 100003, (1 << 20) + 7]


def up(capi, a):
    return capi.Buf.from_numpy(a)


# ----------------------------------------------------------------------------------------------
#  class A: float32 vertical ops
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("op", ["neg", "abs", "sqrt", "floor", "ceil", "round", "trunc", "sin", "cos", "exp", "log",
                                "sign"])
@pytest.mark.parametrize("scale", [1.0, 30.0, 3000.0])
def test_unary_f32_bit_exact(capi, oracle, op, scale):
    a = f32_inputs(100003, seed=11, scale=scale)
    got = capi.unary(op, up(capi, a)).numpy()
    assert bits_equal(got, oracle.unary(op, a)), op


@pytest.mark.parametrize("op", ["tan", "cot", "asin", "acos", "atan", "sinh", "cosh", "tanh", "asinh", "acosh", "atanh",
                                "cbrt"])
@pytest.mark.parametrize("scale", [0.3, 1.0, 30.0, 3000.0])
def test_second_wave_unary_bit_exact(capi, oracle, op, scale):
    """array_math.h second wave as single fused kernels.  The oracle is bit-exact with the reference for the
    functions without rcp() and spells rcp() as an exact division like the kernels do (class C vs the
    reference, tests/test_oracle_vs_ref.py::test_second_wave_class_c), so kernel == oracle bit for bit."""
    a = f32_inputs(100003, seed=31, scale=scale)
    assert bits_equal(capi.unary(op, up(capi, a)).numpy(), oracle.unary(op, a)), op


def test_class_c_kernels_vs_both_rows_of_the_reference(capi):
    """The kernels that go through rcp() / rsqrt() against the REFERENCE (not against our restatement): rcp, rsqrt, division
    bit-exact against its scalar row (1 / a, array_fallbacks.h:23-101); tan, cot, sinh, cosh, tanh, erf, erfc, i0e no further
    from either of its rows than the rows are from each other, within the ulp counts listed in conftest.CLASS_C_BAND
    (fixture: tests/golden/classc_scalar.npz, made by make_golden.py from oracle/_ref/libenoki_refscalar.so + libenoki_ref.so)."""
    from conftest import CLASS_C_BAND, CLASS_C_EXACT, class_c_arg, class_c_check, class_c_fixture
    z = class_c_fixture()
    for op in CLASS_C_EXACT + list(CLASS_C_BAND):
        class_c_check(op, capi.unary(op, up(capi, class_c_arg(op, z))).numpy(), z)
    got = capi.binary("div", up(capi, z["x"]), up(capi, z["y"])).numpy()
    assert bits_equal(got, z["scalar_div"])


@pytest.mark.parametrize("op", ["atan2", "pow", "fmod", "ldexp"])
def test_second_wave_binary_bit_exact(capi, oracle, op):
    for scale in (1.0, 40.0):
        a = f32_inputs(100003, seed=41, scale=scale)
        b = f32_inputs(100003, seed=42, scale=scale)[::-1].copy()
        if op == "ldexp":
            b = np.clip(np.trunc(b), -100, 100).astype(np.float32)
        assert bits_equal(capi.binary(op, up(capi, a), up(capi, b)).numpy(), oracle.binary(op, a, b)), op
        assert bits_equal(capi.binary(op, up(capi, a), 1.5 if op != "ldexp" else 3.0).numpy(),
                          oracle.binary(op, a, np.full_like(a, 1.5 if op != "ldexp" else 3.0))), op


def test_second_wave_golden(capi):
    """kernels against vectors produced by the unmodified reference build (tests/golden/make_golden.py)"""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "elementwise2_f32.npz"))
    a, b = z["in_a"], z["in_b"]
    for op in ["asin", "acos", "atan", "asinh", "acosh", "atanh", "cbrt"]:
        assert bits_equal(capi.unary(op, up(capi, a)).numpy(), z[f"unary_{op}"]), op
    for op in ["atan2", "pow", "fmod"]:
        assert bits_equal(capi.binary(op, up(capi, a), up(capi, b)).numpy(), z[f"binary_{op}"]), op


@pytest.mark.parametrize("scale", [1.0, 30.0, 3000.0, 1e6, 1e300])
def test_f64_transcendentals_bit_exact(capi, oracle, scale):
    """float64 sin/cos/sincos/exp/log: array_math.h double branches, bit-exact vs the oracle (itself bit-exact vs
    the reference build wherever that build is determinate, tests/test_oracle_vs_ref.py)"""
    a = f64_inputs(100003, seed=51, scale=scale)
    for op in ["sin", "cos", "exp", "log"]:
        assert bits_equal(capi.unary(op, up(capi, a)).numpy(), oracle.unary(op, a)), op
    s, c = capi.sincos(up(capi, a))
    es, ec = oracle.sincos(a)
    assert bits_equal(s.numpy(), es) and bits_equal(c.numpy(), ec)
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "elementwise_f64.npz"))
    for op in ["sin", "cos", "exp", "log"]:
        assert bits_equal(capi.unary(op, up(capi, z["in_d"])).numpy(), z[op]), op


@pytest.mark.parametrize("scale", [0.3, 1.0, 30.0, 3000.0])
def test_f64_second_wave_bit_exact(capi, oracle, scale):
    """float64 tan .. cbrt, atan2, pow, fmod, ldexp, sincosh: bit-exact vs the oracle and vs vectors from the reference"""
    a = f64_inputs(100003, seed=61, scale=scale, limit=3e9); b = f64_inputs(100003, seed=62, scale=scale, limit=3e9)[::-1].copy()
    for op in ["tan", "cot", "asin", "acos", "atan", "sinh", "cosh", "tanh", "asinh", "acosh", "atanh", "cbrt"]:
        assert bits_equal(capi.unary(op, up(capi, a)).numpy(), oracle.unary(op, a)), op
    for op in ["atan2", "pow", "fmod"]:
        assert bits_equal(capi.binary(op, up(capi, a), up(capi, b)).numpy(), oracle.binary(op, a, b)), op
    e = np.clip(np.trunc(b), -500, 500)
    assert bits_equal(capi.binary("ldexp", up(capi, a), up(capi, e)).numpy(), oracle.binary("ldexp", a, e))
    s, c = capi.sincosh(up(capi, a))
    assert bits_equal(s.numpy(), oracle.unary("sinh", a)) and bits_equal(c.numpy(), oracle.unary("cosh", a))
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "elementwise_f64.npz"))
    for op in ["tan", "cot", "atan", "sinh", "cosh", "tanh", "asinh", "cbrt"]:
        assert bits_equal(capi.unary(op, up(capi, z["in_d"])).numpy(), z[f"sw_{op}"]), op
    for op in ["atan2", "pow", "fmod"]:
        assert bits_equal(capi.binary(op, up(capi, z["in_d"]), up(capi, z["in_d2"])).numpy(), z[f"sw_{op}"]), op


@pytest.mark.parametrize("n", SIZES)
def test_unary_sizes_and_tails(capi, oracle, n):
    a = f32_inputs(n, seed=n, specials=False)
    for op in ["sin", "exp", "sqrt"]:
        assert bits_equal(capi.unary(op, up(capi, a)).numpy(), oracle.unary(op, a)), (op, n)


def test_sincos_bit_exact(capi, oracle):
    for scale in (1.0, 100.0, 8192.0):
        a = f32_inputs(200001, seed=5, scale=scale)
        s, c = capi.sincos(up(capi, a))
        es, ec = oracle.sincos(a)
        assert bits_equal(s.numpy(), es) and bits_equal(c.numpy(), ec)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_derivative_products_bit_exact(capi, dtype):
    """EK_RCP_SQR / EK_RSQRT_SQR / EK_RSQRT_CUBE -- what the derivatives of rcp and rsqrt are made of (autodiff.h:381-403) as one op
    each -- have the bits of the eager products of the library's own rcp / rsqrt (IEEE division, correctly rounded sqrt)"""
    rng = np.random.default_rng(11)
    a = np.concatenate([rng.uniform(0.01, 100.0, 100003), [0.0, np.inf, 1.0, 4.0, 1e-30, 1e30]]).astype(dtype)
    d = up(capi, a)
    r = capi.unary("rcp", d).numpy(); s = capi.unary("rsqrt", d).numpy()
    with np.errstate(all="ignore"):
        assert bits_equal(capi.unary("rcp_sqr", d).numpy(), r * r)
        assert bits_equal(capi.unary("rsqrt_sqr", d).numpy(), s * s)
        assert bits_equal(capi.unary("rsqrt_cube", d).numpy(), s * (s * s))
    # and applied on load by a reduction (ek_hip_reduce_map): inside the class-D bound of the mapped terms
    b = a[:100003]
    t = (1.0 / b.astype(np.float64)) ** 2
    got = float(capi.reduce_map("hsum", "rcp_sqr", up(capi, b)).numpy()[0])
    assert abs(got - t.sum()) <= (2.0 ** -24 if dtype == np.float32 else 2.0 ** -53) * 64 * np.abs(t).sum()


def test_second_wave_derivative_ops_are_the_compositions(capi):
    """EK_SEC_SQR / EK_SECH_SQR / EK_RCP_1P_SQR (round 6) -- the derivative weights of tan, tanh and atan as one op of the argument each:
    bit for bit sqr(rcp(cos(x))), sqr(rcp(cosh(x))), rcp(1 + sqr(x)) evaluated op by op with the library's own kernels (the compositions the
    reference records, autodiff.h:532-541, 685-696, 606-616), also when a reduction applies them on load"""
    rng = np.random.default_rng(5)
    a = np.concatenate([rng.uniform(-4, 4, 1 << 16), [0.0, -0.0, np.inf, -np.inf, np.nan, 1e-30, 88.0, -88.0, 1.5707964]]).astype(np.float32)
    d = up(capi, a)
    one = up(capi, np.ones_like(a))
    sq = lambda r: capi.binary("mul", r, r)
    want = {"sec_sqr": sq(capi.binary("div", one, capi.unary("cos", d))), "sech_sqr": sq(capi.binary("div", one, capi.unary("cosh", d))),
            "rcp_1p_sqr": capi.binary("div", one, capi.binary("add", one, sq(d)))}
    for op, w in want.items():
        assert bits_equal(capi.unary(op, d).numpy(), w.numpy()), op
    b = rng.uniform(-1.5, 1.5, (1 << 18) + 11).astype(np.float32)
    for op in ("sec_sqr", "sech_sqr", "rcp_1p_sqr", "tanh", "tan", "atan", "sinh", "cosh"):
        got = float(capi.reduce_map("hsum", op, up(capi, b)).numpy()[0])
        ref = float(capi.reduce("hsum", capi.unary(op, up(capi, b))).numpy()[0])
        assert got == ref, (op, got, ref)          # the same reduction tree over the same values



@pytest.mark.parametrize("op", ["add", "sub", "mul", "div", "min", "max", "safe_mul"])
def test_binary_f32_bit_exact(capi, oracle, op):
    a = f32_inputs(100003, seed=1, scale=10.0)
    b = f32_inputs(100003, seed=2, scale=10.0)[::-1].copy()
    got = capi.binary(op, up(capi, a), up(capi, b)).numpy()
    assert bits_equal(got, oracle.binary(op, a, b)), op


@pytest.mark.parametrize("op", ["fmadd", "fmsub", "fnmadd", "fnmsub", "safe_fmadd"])
def test_ternary_f32_bit_exact(capi, oracle, op):
    a = f32_inputs(100003, seed=1); b = f32_inputs(100003, seed=2)[::-1].copy(); c = f32_inputs(100003, seed=3)
    got = capi.ternary(op, up(capi, a), up(capi, b), up(capi, c)).numpy()
    assert bits_equal(got, oracle.ternary(op, a, b, c)), op


def test_denormals_are_not_flushed(capi, oracle):
    """the CPU reference does not set FTZ/DAZ (array_intrin.h:167-194); neither may the kernels"""
    a = np.array([1e-39, 2e-39, -3e-40, 1.17549435e-38, 1e-45] * 13, np.float32)
    b = np.array([0.5, 1.0, 2.0, 0.25, 1.0] * 13, np.float32)
    for op in ["add", "mul", "sub", "safe_mul"]:
        # (safe_mul is ONE v_mul_legacy_f32 on the device: denormal operands and results must behave like the literal
        # compare-and-multiply form of the reference, autodiff.cpp:1191-1199)
        second = b if op in ("mul", "safe_mul") else a
        got = capi.binary(op, up(capi, a), up(capi, second)).numpy()
        assert bits_equal(got, oracle.binary(op, a, second)), op
    z = np.array([0.0, -0.0, 1e-45, -1e-45, 1e-39] * 13, np.float32)             # zeros and denormals against denormals / inf
    w = np.array([1e-45, np.inf, 1e-45, 0.5, np.inf] * 13, np.float32)
    assert bits_equal(capi.binary("safe_mul", up(capi, z), up(capi, w)).numpy(), oracle.binary("safe_mul", z, w))
    got = capi.ternary("fmadd", up(capi, a), up(capi, b), up(capi, a)).numpy()
    assert bits_equal(got, oracle.ternary("fmadd", a, b, a))
    assert np.any((got != 0) & (np.abs(got) < 1.17549435e-38))     # denormal results survive


def test_broadcast_and_immediate_operands(capi, oracle):
    n = 4099
    a = f32_inputs(n, seed=7); x = f32_inputs(n, seed=8); s = np.array([0.75], np.float32)
    full = np.full(n, 0.75, np.float32)
    expect = oracle.ternary("fmadd", a, x, full)
    assert bits_equal(capi.ternary("fmadd", up(capi, a), up(capi, x), up(capi, s)).numpy(), expect)   # size-1 device array
    assert bits_equal(capi.ternary("fmadd", up(capi, a), up(capi, x), 0.75).numpy(), expect)          # immediate
    expect = oracle.binary("safe_mul", full, a)
    assert bits_equal(capi.binary("safe_mul", 0.75, up(capi, a)).numpy(), expect)
    # all operands scalar -> size-1 result computed on the device
    r = capi.binary("mul", up(capi, s), 2.0, n=1).numpy()
    assert r.shape == (1,) and r[0] == np.float32(1.5)


def test_size_mismatch_is_an_error(capi):
    a = up(capi, np.zeros(10, np.float32)); b = up(capi, np.zeros(7, np.float32))
    with pytest.raises(capi.EnokiHipError, match="incompatible size"):
        capi.binary("add", a, b)


def test_misaligned_pointers_take_the_scalar_path(capi, oracle):
    a = f32_inputs(1031, seed=3); b = f32_inputs(1031, seed=4)
    da, db = up(capi, a), up(capi, b)
    for off in (1, 2, 3):
        got = capi.binary("add", da.view(off, 1000), db.view(off, 1000)).numpy()
        assert bits_equal(got, oracle.binary("add", a[off:off + 1000], b[off:off + 1000]))
        got = capi.unary("sin", da.view(off, 1000)).numpy()
        assert bits_equal(got, oracle.unary("sin", a[off:off + 1000]))


# ----------------------------------------------------------------------------------------------
#  class C: rcp / rsqrt against float64 truth with the reference's own bounds
# ----------------------------------------------------------------------------------------------
def test_rcp_rsqrt_vs_f64(capi):
    rng = np.random.default_rng(9)
    a = np.exp(rng.uniform(-80, 80, 200001)).astype(np.float32)
    r = capi.unary("rcp", up(capi, a)).numpy()
    assert ulp_diff(r, (1.0 / a.astype(np.float64)).astype(np.float32)).max() <= 2        # tests/float.cpp:133
    r = capi.unary("rsqrt", up(capi, a)).numpy()
    assert ulp_diff(r, (1.0 / np.sqrt(a.astype(np.float64))).astype(np.float32)).max() <= 3   # tests/float.cpp:152


# ----------------------------------------------------------------------------------------------
#  class A: integers, masks, compares, select, casts
# ----------------------------------------------------------------------------------------------
INT_TYPES = [np.int32, np.uint32, np.int64, np.uint64]


def int_inputs(dt, n, seed):
    rng = np.random.default_rng(seed)
    info = np.iinfo(dt)
    a = rng.integers(info.min, info.max, n, dtype=dt, endpoint=True)
    sp = np.array([0, 1, 2, 3, info.max, info.min, info.max - 1, 31, 32, 33, 63, 64, 65], dtype=dt)
    a[:sp.size] = sp
    return a


@pytest.mark.parametrize("dt", INT_TYPES)
def test_integer_ops_bit_exact(capi, oracle, dt):
    n = 100000
    a, b = int_inputs(dt, n, 1), int_inputs(dt, n, 2)[::-1].copy()
    for op in ["neg", "not", "abs", "popcnt", "lzcnt", "tzcnt"]:
        assert bits_equal(capi.unary(op, up(capi, a)).numpy(), oracle.unary(op, a)), (dt, op)
    bnz = b.copy(); bnz[bnz == 0] = 1
    if np.iinfo(dt).min < 0:
        bnz[bnz == -1] = 3
    for op in ["add", "sub", "mul", "div", "mod", "min", "max", "mulhi", "and", "or", "xor"]:
        bb = bnz if op in ("div", "mod") else b
        assert bits_equal(capi.binary(op, up(capi, a), up(capi, bb)).numpy(), oracle.binary(op, a, bb)), (dt, op)
    bits = 8 * np.dtype(dt).itemsize
    sh = np.random.default_rng(3).integers(0, bits, n).astype(dt)
    for op in ["sl", "sr"]:
        assert bits_equal(capi.binary(op, up(capi, a), up(capi, sh)).numpy(), oracle.binary(op, a, sh)), (dt, op)
    if bits == 32:   # counts >= width: vpsllvd/vpsrlvd/vpsravd semantics (0 / sign fill)
        sh2 = np.random.default_rng(4).integers(0, 40, n).astype(dt)
        for op in ["sl", "sr"]:
            assert bits_equal(capi.binary(op, up(capi, a), up(capi, sh2)).numpy(), oracle.binary(op, a, sh2)), (dt, op)
        for op in ["fmadd", "fmsub", "fnmadd", "fnmsub"]:
            c = a[::-1].copy()
            assert bits_equal(capi.ternary(op, up(capi, a), up(capi, b), up(capi, c)).numpy(),
                              oracle.ternary(op, a, b, c)), (dt, op)


@pytest.mark.parametrize("dt", INT_TYPES + [np.float32, np.float64])
def test_compare_and_select(capi, oracle, dt):
    n = 50021
    if np.dtype(dt).kind == "f":
        a = f32_inputs(n, 5).astype(dt); b = f32_inputs(n, 6).astype(dt); b[::7] = a[::7]
    else:
        a, b = int_inputs(dt, n, 5), int_inputs(dt, n, 6); b[::7] = a[::7]
    for op in ["eq", "neq", "lt", "le", "gt", "ge"]:
        assert np.array_equal(capi.compare(op, up(capi, a), up(capi, b)).numpy(), oracle.compare(op, a, b)), (dt, op)
    m = (np.random.default_rng(1).integers(0, 2, n)).astype(np.uint8)
    assert bits_equal(capi.select(up(capi, m), up(capi, a), up(capi, b)).numpy(), oracle.select(m, a, b))
    # scalar mask / scalar branches
    assert bits_equal(capi.select(True, up(capi, a), up(capi, b)).numpy(), a)
    assert bits_equal(capi.select(up(capi, m), up(capi, a), dt(3)).numpy(), oracle.select(m, a, np.full(n, 3, dt)))


def test_mask_logic(capi):
    rng = np.random.default_rng(2)
    a = rng.integers(0, 2, 10007).astype(np.uint8); b = rng.integers(0, 2, 10007).astype(np.uint8)
    assert np.array_equal(capi.binary("and", up(capi, a), up(capi, b)).numpy(), a & b)
    assert np.array_equal(capi.binary("or", up(capi, a), up(capi, b)).numpy(), a | b)
    assert np.array_equal(capi.binary("xor", up(capi, a), up(capi, b)).numpy(), a ^ b)
    assert np.array_equal(capi.unary("not", up(capi, a)).numpy(), 1 - a)
    assert np.array_equal(capi.compare("eq", up(capi, a), up(capi, b)).numpy(), (a == b).astype(np.uint8))


def test_casts(capi, oracle):
    n = 100000
    f = (np.random.default_rng(1).standard_normal(n) * 1e3).astype(np.float32)
    f[:6] = [0.5, -0.5, 1.5, -1.5, 2.5, -2.5]
    assert bits_equal(capi.cast(up(capi, f), np.int32).numpy(), oracle.cast(f, np.int32))
    assert bits_equal(capi.cast(up(capi, f), np.float64).numpy(), oracle.cast(f, np.float64))
    assert bits_equal(capi.cast(up(capi, np.abs(f)), np.uint32).numpy(), oracle.cast(np.abs(f), np.uint32))
    big = np.array([3e9, -3e9, np.nan, np.inf, -np.inf, 2147483520.0, 2147483648.0, -2147483648.0], np.float32)
    assert bits_equal(capi.cast(up(capi, big), np.int32).numpy(), oracle.cast(big, np.int32))   # cvttps2dq indefinite
    for src in INT_TYPES:
        a = int_inputs(src, n, 3)
        for dst in [np.float32, np.float64] + [t for t in INT_TYPES if t != src]:
            assert bits_equal(capi.cast(up(capi, a), dst).numpy(), oracle.cast(a, dst)), (src, dst)


# ----------------------------------------------------------------------------------------------
#  init ops
# ----------------------------------------------------------------------------------------------
def test_fill_arange_linspace_reverse(capi, oracle):
    assert np.array_equal(capi.fill(np.float32, 2.5, 1001).numpy(), np.full(1001, 2.5, np.float32))
    assert np.array_equal(capi.fill(np.uint8, 1, 77).numpy(), np.ones(77, np.uint8))
    assert np.array_equal(capi.fill(np.int64, -3, 513).numpy(), np.full(513, -3, np.int64))
    assert np.array_equal(capi.arange(np.uint32, 100003).numpy(), np.arange(100003, dtype=np.uint32))
    assert np.array_equal(capi.arange(np.int32, 1000, start=-5, step=3).numpy(), np.arange(-5, 2995, 3, dtype=np.int32))
    assert np.array_equal(capi.arange(np.float32, 5000).numpy(), np.arange(5000, dtype=np.float32))
    # linspace: closed form fmadd(i, step, min) (cuda.h:655-663); the CPU DynamicArray accumulates packet by
    # packet (dynamic.h:924-938), so the two agree to n * eps * range, not bit for bit.
    for n in (2, 7, 1000, 16384):
        got = capi.linspace(np.float32, -1.2, 1.2, n).numpy()
        step = (np.float32(1.2) - np.float32(-1.2)) / np.float32(n - 1)
        expect = (np.arange(n, dtype=np.float64) * np.float64(step) + np.float64(np.float32(-1.2)))
        assert np.abs(got - expect).max() <= 2.4 * 2.0 ** -23
        assert np.abs(got - oracle.linspace(-1.2, 1.2, n)).max() <= 2.4 * n * 2.0 ** -24
    a = f32_inputs(1003, 1)
    assert bits_equal(capi.reverse(up(capi, a)).numpy(), a[::-1].copy())


# ----------------------------------------------------------------------------------------------
#  gather / scatter / scatter_add
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 7, 64, 1000, 100003])
@pytest.mark.parametrize("itype", [np.uint32, np.int32, np.int64, np.uint64])
def test_gather_scatter(capi, oracle, n, itype):
    rng = np.random.default_rng(n)
    K = 257
    src = rng.standard_normal(K).astype(np.float32)
    idx = rng.integers(0, K, n).astype(itype)
    m = (rng.integers(0, 4, n) != 0).astype(np.uint8)
    assert bits_equal(capi.gather(up(capi, src), up(capi, idx), up(capi, m)).numpy(), oracle.gather(src, idx, m))
    assert bits_equal(capi.gather(up(capi, src), up(capi, idx)).numpy(), oracle.gather(src, idx, np.ones(n, np.uint8)))
    # scatter with unique indices (duplicates: last writer in element order on the CPU, unspecified on a GPU)
    perm = rng.permutation(max(K, n))[:n].astype(itype)
    tgt = rng.standard_normal(max(K, n)).astype(np.float32)
    val = rng.standard_normal(n).astype(np.float32)
    d = up(capi, tgt)
    capi.scatter(d, up(capi, val), up(capi, perm), up(capi, m))
    assert bits_equal(d.numpy(), oracle.scatter(tgt, val, perm, m))
    # integer scatter_add is exact regardless of order
    itgt = rng.integers(-100, 100, K).astype(np.int32); ival = rng.integers(-100, 100, n).astype(np.int32)
    d = up(capi, itgt)
    capi.scatter_add(d, up(capi, ival), up(capi, idx), up(capi, m))
    assert np.array_equal(d.numpy(), oracle.scatter(itgt, ival, idx, m, add=True))


def test_gather_other_widths(capi, oracle):
    rng = np.random.default_rng(5)
    n, K = 5003, 99
    idx = rng.integers(0, K, n).astype(np.uint32); m = (rng.integers(0, 3, n) != 0).astype(np.uint8)
    for dt in (np.float64, np.int64, np.uint8):
        src = (rng.integers(0, 2, K).astype(np.uint8) if dt == np.uint8 else (rng.standard_normal(K) * 100).astype(dt))
        expect = np.where(m != 0, src[idx], np.zeros(1, dt)[0]).astype(dt)
        assert bits_equal(capi.gather(up(capi, src), up(capi, idx), up(capi, m)).numpy(), expect), dt


def test_scatter_add_f32_order_bound(capi, oracle):
    """class D: atomics accumulate in unspecified order; bound the error per bin by gamma * sum|v|"""
    rng = np.random.default_rng(6)
    n, K = 200003, 511
    idx = rng.integers(0, K, n).astype(np.uint32); m = (rng.integers(0, 4, n) != 0).astype(np.uint8)
    val = rng.standard_normal(n).astype(np.float32); tgt = rng.standard_normal(K).astype(np.float32)
    d = up(capi, tgt)
    capi.scatter_add(d, up(capi, val), up(capi, idx), up(capi, m))
    got = d.numpy()
    truth = tgt.astype(np.float64).copy(); np.add.at(truth, idx[m != 0], val[m != 0].astype(np.float64))
    mag = np.abs(tgt).astype(np.float64); np.add.at(mag, idx[m != 0], np.abs(val[m != 0]).astype(np.float64))
    cnt = np.bincount(idx[m != 0], minlength=K) + 1
    assert np.all(np.abs(got - truth) <= cnt * 2.0 ** -24 * mag + 1e-30)
    # and the CPU oracle obeys the same bound (sanity of the bound itself)
    assert np.all(np.abs(oracle.scatter(tgt, val, idx, m, add=True) - truth) <= cnt * 2.0 ** -24 * mag + 1e-30)


# ----------------------------------------------------------------------------------------------
#  horizontal reductions
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [0, 1, 2, 7, 8, 9, 255, 256, 1000, 100003, (1 << 22) + 5])
def test_reductions(capi, oracle, n):
    a = f32_inputs(n, seed=n + 1, specials=False)
    for op in ["hsum", "hprod", "hmin", "hmax"]:
        aa = a if op != "hprod" else (1 + 1e-6 * a).astype(np.float32)
        got = capi.reduce(op, up(capi, aa)).numpy()[0]
        if n <= 1 or op in ("hmin", "hmax"):
            assert bits_equal(np.float32(got), np.float32(oracle.reduce(op, aa))), (op, n)   # incl. empty identities
        elif op == "hsum":
            truth = aa.astype(np.float64).sum(); mag = np.abs(aa).astype(np.float64).sum()
            assert abs(got - truth) <= n * 2.0 ** -24 * mag
            assert abs(oracle.reduce(op, aa) - truth) <= n * 2.0 ** -24 * mag
        else:
            truth = np.prod(aa.astype(np.float64))
            assert abs(got - truth) <= n * 2.0 ** -23 * abs(truth)
    for dt in INT_TYPES:
        ia = int_inputs(dt, max(n, 13), 3)[:n]
        for op in ["hsum", "hprod", "hmin", "hmax"]:
            assert capi.reduce(op, up(capi, ia)).numpy()[0] == oracle.reduce(op, ia), (dt, op, n)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n", [2, 9, 1000, 100003, (1 << 20) + 3])
def test_hmin_hmax_nan_and_signed_zero(capi, dtype, n):
    """hmin / hmax are IEEE minNum / maxNum reductions: NaNs are ignored unless every entry is NaN, -0 < +0, whatever
    the reduction order.  (The reference's AVX path is position dependent here -- MINPS returns its second operand
    on unordered / equal compares, dynamic.h:669-702 -- so this is a documented deviation)"""
    rng = np.random.default_rng(n)
    a = rng.standard_normal(n).astype(dtype)
    a[rng.integers(0, n, max(n // 7, 1))] = np.nan
    a[rng.integers(0, n, max(n // 9, 1))] = np.inf
    a[rng.integers(0, n, max(n // 11, 1))] = -np.inf
    cases = {"mixed": a, "first nan": np.concatenate([[np.nan], a[1:]]).astype(dtype), "last nan": np.concatenate([a[:-1], [np.nan]]).astype(dtype),
             "all nan": np.full(n, np.nan, dtype), "zeros": np.where(np.arange(n) % 3 == 0, -0.0, 0.0).astype(dtype),
             "zeros+nan": np.where(np.arange(n) % 2 == 0, np.nan, np.where(np.arange(n) % 3 == 0, -0.0, 0.0)).astype(dtype),
             "finite": rng.standard_normal(n).astype(dtype)}
    for name, v in cases.items():
        lo, hi = capi.reduce("hmin", up(capi, v)).numpy()[0], capi.reduce("hmax", up(capi, v)).numpy()[0]
        with np.errstate(invalid="ignore"):
            elo, ehi = np.fmin.reduce(v), np.fmax.reduce(v)
        zeros = v[v == 0]                       # numpy's fmin does not order the zeros: -0 < +0 here
        if elo == 0 and zeros.size:
            elo = dtype(-0.0) if np.signbit(zeros).any() else dtype(0.0)
        if ehi == 0 and zeros.size:
            ehi = dtype(0.0) if (~np.signbit(zeros)).any() else dtype(-0.0)
        assert bits_equal(np.asarray([lo, hi], dtype), np.asarray([elo, ehi], dtype)), (name, n, lo, hi, elo, ehi)


def test_hsum_run_to_run_deterministic(capi):
    a = up(capi, f32_inputs(1 << 20, 3, specials=False))
    r = {capi.reduce("hsum", a).numpy()[0].tobytes() for _ in range(5)}
    assert len(r) == 1


@pytest.mark.parametrize("n", [0, 1, 15, 16, 17, 1000, 100003])
def test_mask_reductions(capi, oracle, n):
    rng = np.random.default_rng(n)
    for m in (rng.integers(0, 2, n).astype(np.uint8), np.ones(n, np.uint8), np.zeros(n, np.uint8)):
        for op in ["all", "any", "count"]:
            assert capi.mask_reduce(op, up(capi, m)) == oracle.mask_reduce(op, m), (op, n)


def test_hsum_safe_mul_fused(capi):
    n = 100003
    w = f32_inputs(n, 1, specials=False); g = f32_inputs(n, 2, specials=False)
    w[::5] = 0; g[::5] = np.inf                 # 0 * inf must contribute 0, not NaN (autodiff.cpp:1191-1196)
    g[::7] = 0; w[7::35] = np.inf               # inf * 0 likewise
    got = capi.hsum_safe_mul(up(capi, w), up(capi, g)).numpy()[0]
    assert np.isfinite(got)
    w2 = np.where(np.isinf(w), 0, w).astype(np.float32); g2 = np.where(np.isinf(g), 1.0, g).astype(np.float32)
    truth = np.where((w2 == 0) | (g2 == 0) | (w == 0) | (g == 0), 0, w2.astype(np.float64) * g2).sum()
    assert abs(got - truth) <= n * 2.0 ** -24 * np.abs(w2.astype(np.float64) * g2).sum()
    got = capi.hsum_safe_mul(up(capi, w2), up(capi, g2)).numpy()[0]
    truth = np.where((w2 == 0) | (g2 == 0), 0, w2.astype(np.float64) * g2).sum()
    mag = np.abs(w2.astype(np.float64) * g2).sum()
    assert abs(got - truth) <= n * 2.0 ** -24 * mag
    got = capi.hsum_safe_mul(up(capi, w2), 2.0).numpy()[0]                       # immediate gradient
    assert abs(got - 2 * w2.astype(np.float64).sum()) <= n * 2.0 ** -23 * np.abs(w2).sum()
    got = capi.hsum_safe_mul(up(capi, w2), up(capi, np.array([2.0], np.float32)), n=n).numpy()[0]   # device scalar
    assert abs(got - 2 * w2.astype(np.float64).sum()) <= n * 2.0 ** -23 * np.abs(w2).sum()


def test_psum(capi, oracle):
    for n in (1, 5, 256, 257, 4097, 100003):
        a = np.random.default_rng(n).integers(-5, 5, n).astype(np.float32)   # exact in f32 -> order independent
        assert bits_equal(capi.psum(up(capi, a)).numpy(), oracle.psum(a)), n
        ia = np.random.default_rng(n).integers(0, 1000, n).astype(np.uint32)
        assert np.array_equal(capi.psum(up(capi, ia)).numpy(), np.cumsum(ia, dtype=np.uint32)), n


def test_allocator_reuse_and_whos(capi):
    before = capi.lib.ek_hip_launch_count()
    b = capi.Buf(np.float32, 1 << 20); p = b.ptr; b.free()
    b2 = capi.Buf(np.float32, 1 << 20)
    assert b2.ptr == p                      # cached block is reused
    assert "live bytes" in capi.whos()
    assert capi.lib.ek_hip_launch_count() == before


# ----------------------------------------------------------------------------------------------
#  LDS-binned scatter_add (csrc/scatter_binned.hip): large inputs
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("K", [31, 16384, 16385, 100000, 1 << 20, (1 << 22) - 5])
@pytest.mark.parametrize("masked", [False, True])
def test_scatter_add_binned_int_exact(capi, K, masked):
    """integer adds are order independent -> the binned path must reproduce np.add.at exactly"""
    n = (1 << 20) + 1237
    rng = np.random.default_rng(K)
    idx = rng.integers(0, K, n).astype(np.uint32)
    val = rng.integers(-1000, 1000, n).astype(np.int32)
    m = (rng.integers(0, 4, n) != 0).astype(np.uint8) if masked else np.ones(n, np.uint8)
    tgt = rng.integers(-5, 5, K).astype(np.int32)
    d = up(capi, tgt)
    capi.scatter_add(d, up(capi, val), up(capi, idx), up(capi, m) if masked else True)
    expect = tgt.copy(); np.add.at(expect, idx[m != 0], val[m != 0])
    assert np.array_equal(d.numpy(), expect)
    # the two implementations agree with each other
    capi.set_tuning("scatter_add_binned", 0)
    d2 = up(capi, tgt)
    capi.scatter_add(d2, up(capi, val), up(capi, idx), up(capi, m) if masked else True)
    capi.set_tuning("scatter_add_binned", 1)
    assert np.array_equal(d2.numpy(), expect)


@pytest.mark.parametrize("K", [1000, 1 << 20])
def test_scatter_add_binned_f32(capi, K):
    n = (1 << 21) + 5
    rng = np.random.default_rng(K + 1)
    idx = rng.integers(0, K, n).astype(np.int32)
    val = rng.standard_normal(n).astype(np.float32)
    tgt = rng.standard_normal(K).astype(np.float32)
    d = up(capi, tgt)
    capi.scatter_add(d, up(capi, val), up(capi, idx))
    truth = tgt.astype(np.float64); np.add.at(truth, idx, val.astype(np.float64))
    mag = np.abs(tgt).astype(np.float64); np.add.at(mag, idx, np.abs(val).astype(np.float64))
    cnt = np.bincount(idx, minlength=K) + 1
    assert np.all(np.abs(d.numpy() - truth) <= cnt * 2.0 ** -24 * mag + 1e-30)
    # scalar value operand (broadcast gradient)
    d = up(capi, np.zeros(K, np.float32))
    capi.scatter_add(d, 0.5, up(capi, idx), n=n)
    assert np.array_equal(d.numpy(), (np.bincount(idx, minlength=K) * 0.5).astype(np.float32))


# ----------------------------------------------------------------------------------------------
#  ek_hip_scatter_add_multi: several tables through one index array, optional fused edge weights
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("K", [1000, 16385, 1 << 20, (1 << 22) - 3, (1 << 22) + 9])
@pytest.mark.parametrize("count", [1, 2, 3, 4])
def test_scatter_add_multi_int_exact(capi, K, count):
    """integer streams: every table must equal np.add.at exactly, on the fused path (2 .. 256 buckets) and on the
    one-call-per-stream path (tables outside that range)"""
    n = (1 << 19) + 977
    rng = np.random.default_rng(K * 7 + count)
    idx = rng.integers(0, K, n).astype(np.uint32)
    m = (rng.integers(0, 4, n) != 0).astype(np.uint8)
    vals = [rng.integers(-1000, 1000, n).astype(np.int32) for _ in range(count)]
    tgts = [rng.integers(-5, 5, K).astype(np.int32) for _ in range(count)]
    d = [up(capi, t) for t in tgts]
    capi.scatter_add_multi(d, [up(capi, v) for v in vals], up(capi, idx), up(capi, m))
    for c in range(count):
        expect = tgts[c].copy(); np.add.at(expect, idx[m != 0], vals[c][m != 0])
        assert np.array_equal(d[c].numpy(), expect), c


@pytest.mark.parametrize("K", [20000, 1 << 20])
@pytest.mark.parametrize("n", [(1 << 18) + 1, (1 << 21) + 5])
def test_scatter_add_multi_weighted_f32(capi, K, n):
    """float streams with fused weights: stream 0 plain, stream 1 weight array * value array, stream 2 scalar weight *
    value array, stream 3 weight array * scalar value; compared with the exact sum within the rounding bound of an
    arbitrary-order accumulation.  Zero weights / gradients take the safe_mul branch (0 * inf = 0, not NaN)."""
    rng = np.random.default_rng(K + n)
    idx = rng.integers(0, K, n).astype(np.int32)
    g = [rng.standard_normal(n).astype(np.float32) for _ in range(3)]
    w1 = rng.standard_normal(n).astype(np.float32)
    w3 = rng.standard_normal(n).astype(np.float32)
    w1[::7] = 0.0; g[1][::7] = np.inf           # safe_mul: 0 * inf -> 0
    g[2][::11] = 0.0
    tgts = [rng.standard_normal(K).astype(np.float32) for _ in range(4)]
    d = [up(capi, t) for t in tgts]
    capi.scatter_add_multi(d, [up(capi, g[0]), up(capi, g[1]), up(capi, g[2]), 0.75], up(capi, idx),
                           weights=[None, up(capi, w1), -1.5, up(capi, w3)], n=n)
    f32 = np.float32
    with np.errstate(invalid="ignore"):
        prod = [g[0], np.where((w1 == 0) | (g[1] == 0), f32(0), w1 * g[1]).astype(f32), (f32(-1.5) * g[2]).astype(f32),
                (w3 * f32(0.75)).astype(f32)]
    cnt = np.bincount(idx, minlength=K) + 1
    for c in range(4):
        truth = tgts[c].astype(np.float64); np.add.at(truth, idx, prod[c].astype(np.float64))
        mag = np.abs(tgts[c]).astype(np.float64); np.add.at(mag, idx, np.abs(prod[c]).astype(np.float64))
        assert np.all(np.abs(d[c].numpy() - truth) <= cnt * 2.0 ** -24 * mag + 1e-30), c
    # deterministic mode goes through the per-stream path: bit-identical to separate deterministic scatter_adds
    d1 = [up(capi, t) for t in tgts[:2]]
    capi.scatter_add_multi(d1, [up(capi, g[0]), up(capi, g[1])], up(capi, idx), weights=[None, up(capi, w1)], n=n, mode=1)
    d2 = [up(capi, t) for t in tgts[:2]]
    capi.scatter_add(d2[0], up(capi, g[0]), up(capi, idx), mode=1)
    capi.scatter_add(d2[1], up(capi, prod[1]), up(capi, idx), mode=1)
    assert np.array_equal(d1[0].numpy().view(np.uint32), d2[0].numpy().view(np.uint32))
    assert np.array_equal(d1[1].numpy().view(np.uint32), d2[1].numpy().view(np.uint32))


@pytest.mark.parametrize("K", [8193, 100000, (1 << 21) - 3, (1 << 21) + 9])
def test_scatter_add_multi_64bit(capi, K):
    """8-byte element types: 8 Ki bins per LDS bucket (64 KiB), exchange lock on 64-bit words; int64 streams exact, float64
    streams with a fused weight within the rounding bound of an arbitrary-order accumulation"""
    n = (1 << 19) + 977
    rng = np.random.default_rng(K)
    idx = rng.integers(0, K, n).astype(np.uint32)
    m = (rng.integers(0, 4, n) != 0).astype(np.uint8)
    iv = [rng.integers(-10**12, 10**12, n).astype(np.int64) for _ in range(2)]
    it = [rng.integers(-5, 5, K).astype(np.int64) for _ in range(2)]
    d = [up(capi, t) for t in it]
    capi.scatter_add_multi(d, [up(capi, v) for v in iv], up(capi, idx), up(capi, m))
    for c in range(2):
        expect = it[c].copy(); np.add.at(expect, idx[m != 0], iv[c][m != 0])
        assert np.array_equal(d[c].numpy(), expect), c
    g = [rng.standard_normal(n) for _ in range(2)]; w = rng.standard_normal(n); w[::9] = 0.0
    ft = [rng.standard_normal(K) for _ in range(2)]
    d = [up(capi, t) for t in ft]
    capi.scatter_add_multi(d, [up(capi, g[0]), up(capi, g[1])], up(capi, idx), weights=[None, up(capi, w)], n=n)
    prod = [g[0], np.where((w == 0) | (g[1] == 0), 0.0, w * g[1])]
    cnt = np.bincount(idx, minlength=K) + 1
    for c in range(2):
        truth = ft[c].astype(np.longdouble); np.add.at(truth, idx, prod[c].astype(np.longdouble))
        mag = np.abs(ft[c]); np.add.at(mag, idx, np.abs(prod[c]))
        assert np.all(np.abs(d[c].numpy() - truth.astype(np.float64)) <= cnt * 2.0 ** -52 * mag + 1e-300), c


def test_scatter_add_multi_rejects_bad_arguments(capi):
    t = up(capi, np.zeros(100, np.float32)); v = up(capi, np.ones(10, np.float32)); i = up(capi, np.zeros(10, np.uint32))
    with pytest.raises(capi.EnokiHipError):
        capi.scatter_add_multi([t, t], [v, v], i)              # the same table twice
    with pytest.raises(capi.EnokiHipError):
        capi.scatter_add_multi([t] * 5, [v] * 5, i)            # more than 4 streams
    u = up(capi, np.zeros(100, np.float32))
    capi.scatter_add_multi([t, u], [v, 2.0], i, n=10)          # small input: per-stream path
    assert t.numpy()[0] == 10.0 and u.numpy()[0] == 20.0


# ----------------------------------------------------------------------------------------------
#  deterministic scatter_add (mode 1): stable radix sort + sequential per-bin sums == CPU element order
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("K,n", [(7, 1000), (257, 5003), (4096, 100003), (70000, 300001), (1 << 20, (1 << 21) + 11)])
@pytest.mark.parametrize("masked", [False, True])
def test_scatter_add_deterministic_bit_exact(capi, oracle, K, n, masked):
    rng = np.random.default_rng(K + n)
    idx = rng.integers(0, K, n).astype(np.uint32)
    val = (rng.standard_normal(n) * 10.0 ** rng.integers(-3, 4, n)).astype(np.float32)     # wide dynamic range
    m = (rng.integers(0, 4, n) != 0).astype(np.uint8) if masked else np.ones(n, np.uint8)
    tgt = rng.standard_normal(K).astype(np.float32)
    d = up(capi, tgt)
    capi.scatter_add(d, up(capi, val), up(capi, idx), up(capi, m) if masked else True, mode=1)
    assert bits_equal(d.numpy(), oracle.scatter(tgt, val, idx, m, add=True))
    # run-to-run identical as well
    d2 = up(capi, tgt)
    capi.scatter_add(d2, up(capi, val), up(capi, idx.astype(np.int32)), up(capi, m) if masked else True, mode=1)
    assert bits_equal(d2.numpy(), d.numpy())


def test_scatter_add_deterministic_switch(capi, oracle):
    """the global switch (ENOKI_HIP_DETERMINISTIC / tuning) routes the default mode through the sorted path"""
    rng = np.random.default_rng(5)
    K, n = 1000, 200003
    idx = rng.integers(0, K, n).astype(np.uint32); val = rng.standard_normal(n).astype(np.float32)
    tgt = np.zeros(K, np.float32)
    capi.set_tuning("deterministic", 1)
    try:
        d = up(capi, tgt)
        capi.scatter_add(d, up(capi, val), up(capi, idx))
        assert bits_equal(d.numpy(), oracle.scatter(tgt, val, idx, np.ones(n, np.uint8), add=True))
    finally:
        capi.set_tuning("deterministic", 0)


@pytest.mark.parametrize("index_dtype", [np.int64, np.uint64])
@pytest.mark.parametrize("K", [1000, 1 << 20, (1 << 22) + 9])
def test_scatter_add_64bit_index_arrays_take_the_fast_paths(capi, oracle, index_dtype, K):
    """the reference's tape hands the adjoint of a gather 64-bit offsets (autodiff.cpp:355-366): large calls narrow them
    once and run the binned kernels (no device atomics), the multi-table entry and the deterministic sort included"""
    n = (1 << 19) + 977
    rng = np.random.default_rng(K + 3)
    idx = rng.integers(0, K, n).astype(index_dtype)
    m = (rng.integers(0, 4, n) != 0).astype(np.uint8)
    ival = rng.integers(-1000, 1000, n).astype(np.int32)
    tgt = rng.integers(-5, 5, K).astype(np.int32)
    d = up(capi, tgt)
    capi.profile_begin()
    capi.scatter_add(d, up(capi, ival), up(capi, idx), up(capi, m))
    names = {p["kernel"] for p in capi.profile_end()}
    assert "cast" in names and not any(k.startswith("scatter_add_atomic") or k == "scatter_add" for k in names), names
    expect = tgt.copy(); np.add.at(expect, idx[m != 0].astype(np.int64), ival[m != 0])
    assert np.array_equal(d.numpy(), expect)
    # two weighted float tables through one 64-bit index array
    vals = [rng.standard_normal(n).astype(np.float32) for _ in range(2)]
    w = rng.standard_normal(n).astype(np.float32)
    d2 = [up(capi, np.zeros(K, np.float32)) for _ in range(2)]
    capi.scatter_add_multi(d2, [up(capi, v) for v in vals], up(capi, idx), up(capi, m), weights=[up(capi, w), None])
    for c, v in enumerate([vals[0] * w, vals[1]]):
        truth = np.zeros(K); np.add.at(truth, idx[m != 0].astype(np.int64), v[m != 0].astype(np.float64))
        mag = np.zeros(K); np.add.at(mag, idx[m != 0].astype(np.int64), np.abs(v[m != 0]).astype(np.float64))
        cnt = np.bincount(idx[m != 0].astype(np.int64), minlength=K) + 1
        assert np.all(np.abs(d2[c].numpy() - truth) <= cnt * 2.0 ** -24 * mag + 1e-30), c
    # deterministic order: bit-identical to the CPU element order, as with 32-bit indices
    fval = (rng.standard_normal(n) * 10.0 ** rng.integers(-3, 4, n)).astype(np.float32)
    ftgt = rng.standard_normal(K).astype(np.float32)
    d3 = up(capi, ftgt)
    capi.scatter_add(d3, up(capi, fval), up(capi, idx), up(capi, m), mode=1)
    assert bits_equal(d3.numpy(), oracle.scatter(ftgt, fval, idx.astype(np.uint32), m, add=True))


@pytest.mark.parametrize("pattern", ["all_same", "two_bins", "zipf", "one_bucket", "sorted"])
@pytest.mark.parametrize("dt", [np.float32, np.uint32, np.float64, np.int64])
def test_scatter_add_binned_skewed_indices(capi, pattern, dt):
    """heavily skewed index distributions (gradients of a few hot texels): colliding lanes are combined inside the wave
    and the accumulate work is shared out by bucket population, so the result is exact (small integer values) and the
    call does not degenerate (it used to take 0.4 s for 16 Mi adds into one bin)"""
    import time
    n, K = 1 << 22, 1 << 20
    rng = np.random.default_rng(5)
    idx = {"all_same": np.full(n, 12345, np.uint32), "two_bins": ((np.arange(n) % 2) * 700001).astype(np.uint32),
           "zipf": np.minimum(rng.zipf(1.3, n) - 1, K - 1).astype(np.uint32), "one_bucket": rng.integers(0, 16384, n).astype(np.uint32),
           "sorted": np.sort(rng.integers(0, K, n)).astype(np.uint32)}[pattern]
    vals = np.ones(n, dt)
    t = capi.fill(dt, 0, K)
    capi.scatter_add(t, up(capi, vals), up(capi, idx))
    capi.sync()
    t0 = time.perf_counter()
    capi.scatter_add(t, up(capi, vals), up(capi, idx))
    got = t.numpy()
    elapsed = time.perf_counter() - t0
    assert np.array_equal(got.astype(np.int64), 2 * np.bincount(idx, minlength=K))
    assert elapsed < 0.25, elapsed          # incl. the host copies; the degenerate path took > 1 s at this size


@pytest.mark.parametrize("pattern", ["uniform", "zipf", "all_same", "last_slice_only"])
@pytest.mark.parametrize("dt", [np.float32, np.uint32, np.float64, np.uint64])
def test_scatter_add_tables_beyond_4mi_bins(capi, pattern, dt):
    """tables larger than 256 LDS buckets: pairs are first split by 4 Mi-bin slice of the table, every populated slice is
    then an ordinary binned scatter_add (small slices: atomics); exact for small integer values"""
    n, K = 1 << 21, 9_000_001                  # 3 slices, the last one partial
    rng = np.random.default_rng(6)
    idx = {"uniform": rng.integers(0, K, n).astype(np.uint32), "zipf": np.minimum(rng.zipf(1.2, n) - 1, K - 1).astype(np.uint32),
           "all_same": np.full(n, 5_000_000, np.uint32), "last_slice_only": rng.integers(2 << 22, K, n).astype(np.uint32)}[pattern]
    mask = (rng.integers(0, 4, n) != 0).astype(np.uint8)
    vals = rng.integers(1, 4, n).astype(dt)
    t = capi.fill(dt, 1, K)
    capi.scatter_add(t, up(capi, vals), up(capi, idx), up(capi, mask))
    want = 1 + np.bincount(idx[mask != 0], weights=vals[mask != 0].astype(np.float64), minlength=K)
    assert np.array_equal(t.numpy().astype(np.int64), want.astype(np.int64))


def test_out_of_memory_is_an_error_not_a_crash(capi):
    """an allocation beyond the device capacity: the cache is released, the allocation retried once, then the call
    fails with EK_ERR_OOM (the reference exits the process on a failed cudaMalloc, common.cu:268-286 / jit.cu:1716-1723)"""
    small = capi.fill(np.float32, 1.0, 1 << 20)
    with pytest.raises(RuntimeError) as e:
        capi.Buf(np.float32, 1 << 38)                      # 1 TiB
    assert "out of memory" in str(e.value)
    assert float(capi.reduce("hsum", small).numpy()[0]) == float(1 << 20)      # the library keeps working


@pytest.mark.parametrize("seed", range(48))
def test_scatter_add_randomized_shapes(capi, seed):
    """random (n, table size, mask density, index distribution, value type) through the size-dependent paths of
    ek_hip_scatter_add (atomics / LDS direct / binned / two-level binned); small integer values make every path exact"""
    rng = np.random.default_rng(100 + seed)
    n = int(rng.choice([1, 7, 1000, (1 << 18) - 1, 1 << 18, (1 << 18) + 1, 300_007, 1 << 20, (1 << 21) + 12345]))
    K = int(rng.choice([1, 2, 63, 1024, 1025, 16384, 16385, 100_000, (1 << 20) - 3, 1 << 22, (1 << 22) + 1, 5_000_011, 1 << 24]))
    dt = [np.float32, np.uint32, np.int32, np.float64, np.int64, np.uint64][seed % 6]
    kind = ["uniform", "clustered", "few", "ramp"][int(rng.integers(0, 4))]
    if kind == "uniform":
        idx = rng.integers(0, K, n)
    elif kind == "clustered":
        idx = np.clip(rng.normal(K * 0.7, max(K * 0.01, 1.0), n), 0, K - 1).astype(np.int64)
    elif kind == "few":
        idx = rng.choice(rng.integers(0, K, 5), n)
    else:
        idx = (np.arange(n) * 977) % K
    idx = idx.astype(np.uint32)
    density = float(rng.choice([0.0, 0.3, 1.0]))
    mask = (rng.random(n) < density).astype(np.uint8)
    vals = rng.integers(0, 3, n).astype(dt)
    t = capi.fill(dt, 2, K)
    use_mask = density < 1.0
    if use_mask:
        capi.scatter_add(t, up(capi, vals), up(capi, idx), up(capi, mask))
    else:
        capi.scatter_add(t, up(capi, vals), up(capi, idx))
    sel = mask != 0 if use_mask else np.ones(n, bool)
    want = 2 + np.bincount(idx[sel], weights=vals[sel].astype(np.float64), minlength=K)
    assert np.array_equal(t.numpy().astype(np.int64), want.astype(np.int64)), (n, K, dt, kind, density)


def test_concat_single_launch(capi):
    """ek_hip_concat: several arrays -> one flat buffer in one launch (staging of the packed all-reduce)"""
    import ctypes
    rng = np.random.default_rng(3)
    parts = [rng.standard_normal(s).astype(np.float32) for s in (1, 1000, 0, 70001, 3)]
    bufs = [up(capi, p) if p.size else None for p in parts]
    out = capi.Buf(np.float32, sum(p.size for p in parts))
    srcs = (ctypes.c_void_p * len(parts))(*[b.ptr if b is not None else None for b in bufs])
    sizes = (ctypes.c_size_t * len(parts))(*[p.size for p in parts])
    before = capi.lib.ek_hip_launch_count()
    capi.check(capi.lib.ek_hip_concat(capi.NP2EK[np.dtype(np.float32)], ctypes.c_void_p(out.ptr), len(parts), srcs, sizes))
    assert capi.lib.ek_hip_launch_count() - before == 1
    assert np.array_equal(out.numpy(), np.concatenate(parts))
    with pytest.raises(capi.EnokiHipError):
        capi.check(capi.lib.ek_hip_concat(capi.NP2EK[np.dtype(np.float32)], ctypes.c_void_p(out.ptr), 9, srcs, sizes))
