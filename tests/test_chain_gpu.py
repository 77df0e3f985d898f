"""Chains: base(src..) under up to three unary maps in ONE pass (ek_hip_reduce_chain / ek_hip_map_chain, csrc/reduce.hip) and the
HIPArray nodes that feed them (include/enoki/hip.h: kind 4 = an unevaluated fma / product / product-then-sum over evaluated
operands, kind 1 maps on top of it).  What the reference's JIT does for hsum(sin(exp(fmadd(a, x, b)))) -- BASELINE configs[1] --
one kernel over a, x, b (src/cuda/jit.cu:1066-1217, :1418-1508).

Yardsticks: the values of a chain are BIT-IDENTICAL to the op-by-op kernels of the same library (which are pinned bit for bit
against the reference build in test_kernels_gpu.py) and to the CPU oracle; reductions over them are class D (order of fp
additions) and are held to the float64 sum of the f32 terms; hmin / hmax are order independent: bit for bit."""
import json

import numpy as np
import pytest

from conftest import bits_equal, f32_inputs, f64_inputs, uniform_pm1

pytestmark = pytest.mark.gpu
EPS = {np.float32: 2.0 ** -24, np.float64: 2.0 ** -53}


def up(capi, a):
    return capi.Buf.from_numpy(a)


BASES = [("fmadd", 3), ("fmsub", 3), ("fnmadd", 3), ("fnmsub", 3), ("muladd", 3), ("mulsub", 3), ("nmuladd", 3),
         ("add", 2), ("sub", 2), ("mul", 2), (None, 1)]
MAPS = [[], ["sin"], ["exp", "sin"], ["abs", "sqrt", "rcp"], ["neg", "exp", "log"], ["cos", "abs", "rsqrt"], ["rcp_sqr"], ["abs", "rsqrt_sqr", "rsqrt_cube"],
        # round 6: the second-wave functions whose derivative is one map of the argument, and those derivative maps
        ["tanh"], ["sech_sqr"], ["tan", "abs"], ["sec_sqr"], ["atan", "sinh"], ["rcp_1p_sqr", "cosh"]]


def op_by_op(capi, base, srcs, maps):
    if len(srcs) == 3:
        if base in ("muladd", "mulsub", "nmuladd"):            # the operator spelling: a product, then a sum
            p = capi.binary("mul", srcs[0], srcs[1])
            v = capi.binary("add", p, srcs[2]) if base == "muladd" else capi.binary("sub", p, srcs[2]) if base == "mulsub" else capi.binary("sub", srcs[2], p)
        else:
            v = capi.ternary(base, *srcs)
    elif len(srcs) == 2:
        v = capi.binary(base, *srcs)
    else:
        v = srcs[0]
    for m in maps:
        v = capi.unary(m, v)
    return v


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("base,arity", BASES)
def test_chain_values_are_the_op_by_op_values(capi, dtype, base, arity):
    """every base under every map list, a ragged length (vector body + guarded tail), arrays / a one-element array / an immediate"""
    n = 100003
    gen = f32_inputs if dtype == np.float32 else f64_inputs
    arrs = [up(capi, gen(n, seed=3 + k, scale=2.0)) for k in range(3)]
    one = up(capi, np.array([0.75], dtype))
    for maps in MAPS:
        for variant in range(3):
            srcs = list(arrs[:arity])
            if variant == 1 and arity >= 2:
                srcs[1] = 0.625                                  # host scalar
            if variant == 2 and arity == 3:
                srcs[2] = one                                    # device scalar
            if variant and arity == 1:
                continue
            want = op_by_op(capi, base, srcs, maps).numpy()
            got = capi.map_chain(base, srcs, maps).numpy()
            assert bits_equal(got, want), (base, maps, variant)
            for rop in ("hmin", "hmax"):
                r = capi.reduce_chain(rop, base, srcs, maps).numpy()[0]
                finite = want[~np.isnan(want)]
                if finite.size:
                    assert bits_equal(np.array([r]), np.array([finite.min() if rop == "hmin" else finite.max()])), (base, maps, rop)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_chain_sums_are_class_d(capi, dtype):
    n = (1 << 20) + 77
    a, x, b = (uniform_pm1(n, s).astype(dtype) for s in (1, 2, 3))
    da, dx, db = up(capi, a), up(capi, x), up(capi, b)
    for base, srcs, maps in (("fmadd", [da, dx, db], ["exp", "sin"]), ("muladd", [da, dx, db], ["sin"]), ("fmadd", [da, dx, db], []),
                             ("mul", [da, dx], ["cos"]), (None, [da], ["abs", "sqrt", "exp"])):
        terms = op_by_op(capi, base, srcs, maps).numpy().astype(np.float64)
        got = float(capi.reduce_chain("hsum", base, srcs, maps).numpy()[0])
        depth = n // (1 << 18) + 40
        assert abs(got - terms.sum()) <= EPS[dtype] * depth * np.abs(terms).sum(), (base, maps)
        assert abs(got - terms.sum()) <= EPS[dtype] * (8 * np.sqrt(depth * (terms ** 2).sum()) + 4 * abs(terms.sum())), (base, maps)
    small = up(capi, (1.0 + 0.05 * a[:4099]).astype(dtype))            # factors near 1: the product of 4099 of them stays in range
    sc = float(np.asarray(1.001, dtype))
    p64 = float(np.exp(np.log(small.numpy().astype(np.float64) * sc).sum()))
    got = float(capi.reduce_chain("hprod", "mul", [small, sc], []).numpy()[0])
    assert abs(got - p64) <= 4 * 4099 * EPS[dtype] * abs(p64) + 1e-300


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_chain_with_a_factor_and_a_second_output(capi, dtype):
    """ek_hip_map_chain_product: (chain * scale, op2(w, chain * scale)) -- bit for bit the op-by-op kernels, for both products, with
    zeros in w against infinities / NaNs in the chain (safe_mul's point), a ragged length, either output alone"""
    n = 70001
    gen = f32_inputs if dtype == np.float32 else f64_inputs
    arrs = [up(capi, gen(n, seed=11 + k, scale=2.0)) for k in range(3)]
    wn = gen(n, seed=17, scale=2.0)
    wn[::5] = 0.0
    wn[1::7] = -0.0
    w = up(capi, wn)
    for base, arity in (("fmadd", 3), ("nmuladd", 3), ("mul", 2), (None, 1)):
        for maps in ([], ["cos"], ["exp", "sin"], ["log"], ["neg", "rcp", "sqrt"]):
            if not maps and arity == 1:
                continue
            srcs = list(arrs[:arity])
            for scale in (None, -1.0, 0.375):
                v = op_by_op(capi, base, srcs, maps)
                if scale is not None:
                    v = capi.binary("mul", v, dtype(scale))
                for op2 in ("mul", "safe_mul"):
                    want2 = capi.binary(op2, w, v).numpy()
                    got, got2 = capi.map_chain_product(base, srcs, maps, scale=scale, w=w, op2=op2)
                    assert bits_equal(got.numpy(), v.numpy()), (base, maps, scale)
                    assert bits_equal(got2.numpy(), want2), (base, maps, scale, op2)
                _, only2 = capi.map_chain_product(base, srcs, maps, scale=scale, w=w, op2="safe_mul", first=False)
                assert bits_equal(only2.numpy(), capi.binary("safe_mul", w, v).numpy()), (base, maps, scale)
                only, none = capi.map_chain_product(base, srcs, maps, scale=scale)
                assert none is None and bits_equal(only.numpy(), v.numpy()), (base, maps, scale)


@pytest.fixture(scope="module")
def ek():
    import enoki_amd.hip as m
    m.hip_init(0)
    return m


def kernels(m, fn):
    m.hip_profile_begin()
    out = fn()
    prof = json.loads(m.hip_profile_end())
    return out, {k["kernel"]: k["launches"] for k in prof if k["launches"]}


def test_cfg2_expression_is_one_pass(ek, oracle):
    """BASELINE configs[1]: hsum(sin(exp(fmadd(a, x, b)))) on plain arrays -- one chain reduction (+ its second stage), nothing
    written; the same expression with the operands kept alive afterwards, written with operators, and forced element by element
    gives the same bits as the oracle's op-by-op evaluation"""
    n = (1 << 20) + 13
    a, x, b = uniform_pm1(n, 1), uniform_pm1(n, 2), uniform_pm1(n, 3)
    da, dx, db = ek.Float32(a), ek.Float32(x), ek.Float32(b)
    u = oracle.ternary("fmadd", a, x, b)
    t = oracle.unary("sin", oracle.unary("exp", u)).astype(np.float64)
    depth = n // (1 << 18) + 40
    y, ks = kernels(ek, lambda: float(ek.hsum(ek.sin(ek.exp(ek.fmadd(da, dx, db)))).numpy()[0]))
    assert set(ks) == {"reduce_chain", "reduce_stage2"} and ks["reduce_chain"] == 1, ks
    assert abs(y - t.sum()) <= 2.0 ** -24 * depth * np.abs(t).sum()
    # written with operators: a product and a sum, two roundings -- still one pass
    t2 = oracle.unary("sin", oracle.binary("add", oracle.binary("mul", a, x), b)).astype(np.float64)
    y2, ks = kernels(ek, lambda: float(ek.hsum(ek.sin(da * dx + db)).numpy()[0]))
    assert set(ks) == {"reduce_chain", "reduce_stage2"}, ks
    assert abs(y2 - t2.sum()) <= 2.0 ** -24 * depth * np.abs(t2).sum()
    y3, ks = kernels(ek, lambda: float(ek.hsum(db - da * dx).numpy()[0]))
    assert set(ks) == {"reduce_chain", "reduce_stage2"}, ks
    t3 = oracle.binary("sub", b, oracle.binary("mul", a, x)).astype(np.float64)
    assert abs(y3 - t3.sum()) <= 2.0 ** -24 * depth * np.abs(t3).sum()
    # forced: the chain is written by ONE kernel, bit-identical to the oracle's op-by-op values
    v, ks = kernels(ek, lambda: ek.sin(ek.exp(ek.fmadd(da, dx, db))).numpy())
    assert ks.get("map_chain") == 1 and not any(k in ks for k in ("fmadd", "exp", "sin")), ks
    assert bits_equal(v, oracle.unary("sin", oracle.unary("exp", u)))
    w = (da * dx + db).numpy()
    assert bits_equal(w, oracle.binary("add", oracle.binary("mul", a, x), b))


def test_a_value_somebody_else_holds_is_evaluated_once(ek, oracle):
    """the chain only absorbs what nobody else wants: u held by the caller is written once and then reused; results are the same"""
    n = (1 << 18) + 5
    a, x, b = uniform_pm1(n, 1), uniform_pm1(n, 2), uniform_pm1(n, 3)
    da, dx, db = ek.Float32(a), ek.Float32(x), ek.Float32(b)
    u = ek.fmadd(da, dx, db)
    e = ek.exp(u)
    y1 = float(ek.hsum(ek.sin(e)).numpy()[0])               # e and u are held: evaluated, sin applied on load
    y2 = float(ek.hsum(ek.cos(e)).numpy()[0])
    assert bits_equal(u.numpy(), oracle.ternary("fmadd", a, x, b))
    assert bits_equal(e.numpy(), oracle.unary("exp", oracle.ternary("fmadd", a, x, b)))
    t = oracle.unary("exp", oracle.ternary("fmadd", a, x, b))
    s64, c64 = oracle.unary("sin", t).astype(np.float64), oracle.unary("cos", t).astype(np.float64)
    assert abs(y1 - s64.sum()) <= 2.0 ** -24 * 48 * np.abs(s64).sum() and abs(y2 - c64.sum()) <= 2.0 ** -24 * 48 * np.abs(c64).sum()
    # an operand that is overwritten AFTER the node was made: the node sees the old contents (arrays are values)
    p = da * dx
    ek.scatter(da, ek.Float32(np.zeros(16, np.float32)), ek.UInt32(np.arange(16, dtype=np.uint32)))
    assert bits_equal(p.numpy(), oracle.binary("mul", a, x))


def test_explain_and_the_bucket_order_log(ek, capfd):
    """array.explain() names the state of an array without evaluating it; hip_set_log_level(2) prints one line when an expression that
    could have run in bucket order is evaluated in element order, and why"""
    import enoki_amd.hip_autodiff as ad
    n, K = 1 << 19, 1 << 16
    rng = np.random.default_rng(5)
    A, B = ek.Float32(uniform_pm1(K, 1)), ek.Float32(uniform_pm1(K, 2))
    x = ek.Float32(uniform_pm1(n, 3))
    idx = ek.UInt32(rng.integers(0, K, n).astype(np.uint32))
    launches = ek.hip_launch_count()
    g = ek.gather(A, idx)
    assert "unevaluated gather" in g.explain()
    p = g * x
    assert "product of a gather" in p.explain() and "BUCKET ORDER possible" in p.explain()
    u = p + ek.gather(B, idx)
    assert "product-then-sum of two gathers" in u.explain()
    s = ek.sin(u)
    assert "unevaluated unary op" in s.explain() and "kind 2" in s.explain()
    f = ek.fmadd(x, x, x)
    assert "unevaluated arithmetic op" in f.explain()
    assert ek.hip_launch_count() == launches, "explain() must not evaluate anything"
    assert "evaluated array" in x.explain() and "host scalar" in ek.Float32(1.5).explain()
    # an elementwise consumer of the node: element order, and the log says so
    ek.hip_set_log_level(2)
    try:
        capfd.readouterr()
        v = (u * ek.Float32(2.0)).numpy()
        err = capfd.readouterr().err
    finally:
        ek.hip_set_log_level(0)
    assert "[bucket order]" in err and "ELEMENT order" in err, err
    assert "evaluated array" in u.explain()
    want = (uniform_pm1(K, 1)[idx.numpy()] * uniform_pm1(n, 3) + uniform_pm1(K, 2)[idx.numpy()]) * np.float32(2.0)
    assert bits_equal(v, want.astype(np.float32))


def test_cfg3a_is_a_reduction_and_one_backward_pass(ek, oracle):
    """BASELINE configs[2] with leaf arrays: y = hsum(sin(fmadd(a, x, b))), backward().  Forward: one chain reduction over a, x, b
    (u is wanted by the sin and by the cos that differentiating sin records, by nobody else: never written).  Backward: grad_b =
    cos(u) and grad_a = safe_mul(x, cos(u)) are the two outputs of ONE pass over a, x, b: 32 B/elt for the step.  Gradients are
    vertical ops: bit for bit the oracle's (class A)."""
    import enoki_amd.hip_autodiff as ad
    n = (1 << 20) + 29
    a, x, b = uniform_pm1(n, 1), uniform_pm1(n, 2), uniform_pm1(n, 3)
    x[::9] = 0.0
    xd = ad.Float32(x)

    def step():
        da, db = ad.Float32(a), ad.Float32(b)
        ad.set_requires_gradient(da); ad.set_requires_gradient(db)
        y = ad.hsum(ad.sin(ad.fmadd(da, xd, db)))
        ad.backward(y)
        return float(ad.detach(y).numpy()[0]), ad.gradient(da), ad.gradient(db)

    step()
    (y, ga, gb), ks = kernels(ek, lambda: step())
    ga, gb = ga.numpy(), gb.numpy()
    big = {k: v for k, v in ks.items() if k not in ("reduce_stage2", "copy", "memcpy", "fill")}
    assert big == {"reduce_chain": 1, "map_chain_product": 1}, ks
    u = oracle.ternary("fmadd", a, x, b)
    c = oracle.unary("cos", u)
    assert bits_equal(gb, c)
    assert bits_equal(ga, oracle.binary("safe_mul", x, c))
    t = oracle.unary("sin", u).astype(np.float64)
    assert abs(y - t.sum()) <= 2.0 ** -24 * (n // (1 << 18) + 40) * np.abs(t).sum()


@pytest.mark.parametrize("fn", ["tanh", "tan", "atan", "sinh", "cosh"])
def test_second_wave_maps_over_leaf_arrays_are_two_passes(ek, oracle, fn):
    """round 6: tan, tanh, atan, sinh, cosh with gradients stay a chain -- the value is an unevaluated map of u and so is the weight
    the tape records (EK_SEC_SQR = sqr(sec(u)), EK_SECH_SQR = sqr(sech(u)), EK_RCP_1P_SQR = rcp(1 + sqr(u)), cosh(u), sinh(u): one op
    of the argument each, with the roundings of the reference's compositions, autodiff.h:532-541, 606-616, 635-657, 685-696), so
    forward + backward() are two bandwidth kernels; the weights are bit for bit the op-by-op compositions of the library's own
    kernels (rcp is the exact division: class C against the reference's AVX rows, like every rcp-based weight)"""
    import enoki_amd.hip_autodiff as ad
    import enoki_amd.hip as ekc
    n = (1 << 19) + 3
    a, x, b = uniform_pm1(n, 4), uniform_pm1(n, 5), uniform_pm1(n, 6)
    xd = ad.Float32(x)
    f = {"tanh": ad.tanh, "tan": ad.tan, "atan": ad.atan, "sinh": ad.sinh, "cosh": ad.cosh}[fn]

    def step():
        da, db = ad.Float32(a), ad.Float32(b)
        ad.set_requires_gradient(da); ad.set_requires_gradient(db)
        y = ad.hsum(f(ad.fmadd(da, xd, db)))
        ad.backward(y)
        return float(ad.detach(y).numpy()[0]), ad.gradient(da), ad.gradient(db)

    step()
    (y, ga, gb), ks = kernels(ek, lambda: step())
    ga, gb = ga.numpy(), gb.numpy()
    big = {k: v for k, v in ks.items() if k not in ("reduce_stage2", "copy", "memcpy", "fill")}
    assert big == {"reduce_chain": 1, "map_chain_product": 1}, (fn, ks)
    # op by op with the library's own kernels (evaluated eagerly on small pieces of the same data: below the deferral threshold)
    U = ekc.Float32
    ekc.hip_set_defer(False)
    try:
        u = ekc.fmadd(U(a), U(x), U(b))
        one = U(np.ones(n, np.float32))
        w = {"tanh": lambda: (lambda r: r * r)(one / ekc.cosh(u)), "tan": lambda: (lambda r: r * r)(one / ekc.cos(u)),
             "atan": lambda: one / (one + u * u), "sinh": lambda: ekc.cosh(u), "cosh": lambda: ekc.sinh(u)}[fn]()
        v = {"tanh": ekc.tanh, "tan": ekc.tan, "atan": ekc.atan, "sinh": ekc.sinh, "cosh": ekc.cosh}[fn](u)
        w_np, v_np = w.numpy(), v.numpy().astype(np.float64)
    finally:
        ekc.hip_set_defer(True)
    assert bits_equal(gb, w_np), fn
    assert bits_equal(ga, oracle.binary("safe_mul", x, w_np)), fn
    assert abs(y - v_np.sum()) <= 2.0 ** -24 * (n // (1 << 18) + 40) * np.abs(v_np).sum()


@pytest.mark.parametrize("fn", ["cos", "exp", "log_abs"])
def test_other_maps_over_leaf_arrays_are_two_passes_too(ek, oracle, fn):
    """the same two passes for the other differentiable maps whose derivative is a fusable map of u: cos (weight -sin(u): a scaled
    sibling), exp (the weight IS the result), log (weight rcp(u)); gradients bit for bit"""
    import enoki_amd.hip_autodiff as ad
    n = (1 << 19) + 3
    a, x, b = uniform_pm1(n, 4), uniform_pm1(n, 5), uniform_pm1(n, 6)
    if fn == "log_abs":
        b = (b + np.float32(3.0)).astype(np.float32)              # u in [1, 5]
    xd = ad.Float32(x)
    f = {"cos": ad.cos, "exp": ad.exp, "log_abs": ad.log}[fn]

    def step():
        da, db = ad.Float32(a), ad.Float32(b)
        ad.set_requires_gradient(da); ad.set_requires_gradient(db)
        y = ad.hsum(f(ad.fmadd(da, xd, db)))
        ad.backward(y)
        return ad.gradient(da), ad.gradient(db)

    step()
    (ga, gb), ks = kernels(ek, lambda: step())
    ga, gb = ga.numpy(), gb.numpy()
    big = {k: v for k, v in ks.items() if k not in ("reduce_stage2", "copy", "memcpy", "fill")}
    assert big == {"reduce_chain": 1, "map_chain_product": 1}, (fn, ks)
    u = oracle.ternary("fmadd", a, x, b)
    w = {"cos": lambda: oracle.unary("neg", oracle.unary("sin", u)), "exp": lambda: oracle.unary("exp", u),
         "log_abs": lambda: oracle.unary("rcp", u)}[fn]()
    assert bits_equal(gb, w), fn
    assert bits_equal(ga, oracle.binary("safe_mul", x, w)), fn
