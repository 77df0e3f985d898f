"""Unary operations applied on load (ek_hip_reduce_map, ek_hip_scatter_add_multi_map) and the deferred unary results of
HIPArray that feed them.  The map is the SAME device function as the stand-alone kernel and the reductions keep their
tree, so every result here is bit-identical to "evaluate, then consume" (class A relative to the unfused path)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import bits_equal, f32_inputs, f64_inputs

pytestmark = pytest.mark.gpu

MAP_OPS = ["neg", "abs", "sqrt", "rcp", "rsqrt", "sin", "cos", "exp", "log"]
SIZES = [1, 5, 255, 1024, 4099, 100003, (1 << 20) + 7]


def up(capi, a):
    return capi.Buf.from_numpy(a)


def _inputs(dtype, n, seed, op):
    gen = f32_inputs if dtype == np.float32 else f64_inputs
    x = gen(n, seed=seed).astype(dtype)
    if op in ("sqrt", "rsqrt", "log"):
        x = np.abs(x) + dtype(1e-3)
    if op == "exp":
        x = np.clip(x, -20, 20).astype(dtype)
    return x


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("op", MAP_OPS)
def test_reduce_map_equals_unary_then_reduce(capi, dtype, op):
    for n in SIZES:
        x = _inputs(dtype, n, n + 11, op)
        dx = up(capi, x)
        mapped = capi.unary(op, dx)
        for red in ("hsum", "hmax", "hmin", "hprod"):
            if red == "hprod" and n > 5000:
                continue
            ref = capi.reduce(red, mapped)
            got = capi.reduce_map(red, op, dx)
            assert bits_equal(got.numpy(), ref.numpy()), (op, red, n)


def test_reduce_map_rejects_unfusable(capi):
    dx = up(capi, np.ones(100, np.float32))
    with pytest.raises(Exception):
        capi.reduce_map("hsum", "asinh", dx)             # (tanh, tan, atan, sinh, cosh ARE applied on load since round 6)
    assert float(capi.reduce_map("hsum", "tanh", dx).numpy()[0]) == float(capi.reduce("hsum", capi.unary("tanh", dx)).numpy()[0])
    with pytest.raises(Exception):
        capi.reduce_map("hsum", "sin", up(capi, np.ones(100, np.uint32)))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_scatter_add_multi_map(capi, dtype):
    """value streams mapped on load == streams evaluated first; every path: LDS-binned (large), small table (direct),
    deterministic (mode 1: bit-exact in element order)"""
    rng = np.random.default_rng(3)
    for n, K in ((200003, 1 << 17), (1 << 20, 1 << 18), (50000, 1000), (70000, 1 << 17)):
        u = (rng.standard_normal(n) * 3).astype(dtype)       # finite values: the unordered sums are compared by bound
        w = rng.standard_normal(n).astype(dtype)
        idx = rng.integers(0, K, n).astype(np.uint32)
        du, dw, di = up(capi, u), up(capi, w), up(capi, idx)
        c = capi.unary("cos", du)
        for mode in (1, 0):
            ref = [up(capi, np.zeros(K, dtype)) for _ in range(2)]
            got = [up(capi, np.zeros(K, dtype)) for _ in range(2)]
            capi.scatter_add_multi(ref, [c, c], di, weights=[dw, None], mode=mode)
            capi.scatter_add_multi_map(got, [du, du], ["cos", "cos"], di, weights=[dw, None], mode=mode)
            for r, g in zip(ref, got):
                if mode == 1:
                    assert bits_equal(g.numpy(), r.numpy()), (n, K)
                else:                                  # unordered accumulation on both sides: same addends per bin
                    terms = np.abs(c.numpy().astype(np.float64)) * (np.abs(w.astype(np.float64)) if r is ref[0] else 1.0)
                    bound = np.bincount(idx, minlength=K) * np.bincount(idx, weights=terms, minlength=K) * np.finfo(dtype).eps
                    assert np.all(np.abs(g.numpy().astype(np.float64) - r.numpy()) <= bound), (n, K)
        # different ops per stream, one stream unmapped, three streams
        s = capi.unary("sin", du)
        ref = [up(capi, np.zeros(K, dtype)) for _ in range(3)]
        got = [up(capi, np.zeros(K, dtype)) for _ in range(3)]
        capi.scatter_add_multi(ref, [s, dw, c], di, weights=[None, None, dw], mode=1)
        capi.scatter_add_multi_map(got, [du, dw, du], ["sin", None, "cos"], di, weights=[None, None, dw], mode=1)
        for r, g in zip(ref, got):
            assert bits_equal(g.numpy(), r.numpy())


def test_scatter_add_multi_map_integer_counts(capi):
    """the binned path itself (mode 0) with a mapped stream: exp(0) = 1 per element -> exact counts"""
    rng = np.random.default_rng(9)
    n, K = 1 << 21, 1 << 18
    idx = rng.integers(0, K, n).astype(np.uint32)
    dz, di = up(capi, np.zeros(n, np.float32)), up(capi, idx)
    t0, t1 = up(capi, np.zeros(K, np.float32)), up(capi, np.zeros(K, np.float32))
    capi.scatter_add_multi_map([t0, t1], [dz, dz], ["exp", "cos"], di)
    cnt = np.bincount(idx, minlength=K).astype(np.float32)
    assert np.array_equal(t0.numpy(), cnt) and np.array_equal(t1.numpy(), cnt)


# ----------------------------------------------------------------------------------------------
#  HIPArray level: deferred unary results keep value semantics
# ----------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ek():
    import enoki_amd.hip as m
    m.hip_init(0)
    return m


def test_deferred_map_semantics(ek):
    n = 300000
    x = f32_inputs(n, seed=21)
    dx = ek.Float32(x)
    eager = {}
    ek.hip_set_defer_gather(False)
    try:
        eager["sin"] = ek.sin(dx).numpy(); eager["hsum_sin"] = ek.hsum(ek.sin(dx)).numpy()
        s0, c0 = ek.sincos(dx); eager["cos"] = c0.numpy()
        eager["hmax_abs"] = ek.hmax(ek.abs(dx)).numpy()
    finally:
        ek.hip_set_defer_gather(True)

    # consumed by a reduction: one pass, nothing materialised
    l0 = ek.hip_launch_count()
    y = ek.hsum(ek.sin(dx))
    assert ek.hip_launch_count() - l0 == 2                     # reduce stage 1 + 2; no sin kernel
    assert bits_equal(y.numpy(), eager["hsum_sin"])
    assert bits_equal(ek.hmax(ek.abs(dx)).numpy(), eager["hmax_abs"])
    # looked at directly: an ordinary array
    s = ek.sin(dx)
    assert bits_equal(s.numpy(), eager["sin"])
    # consumed by an elementwise op: materialised first, same bits
    s = ek.sin(dx)
    assert bits_equal((s + dx).numpy(), eager["sin"] + x)
    # reduced AND used afterwards
    s = ek.sin(dx)
    y = ek.hsum(s)
    assert bits_equal(y.numpy(), eager["hsum_sin"]) and bits_equal(s.numpy(), eager["sin"])
    # the source changes after the op: the result must not
    T = ek.Float32(x)
    s = ek.sin(T)
    ek.scatter(T, ek.Float32(np.zeros(n, np.float32)), ek.UInt32.arange(n))
    assert bits_equal(s.numpy(), eager["sin"]) and np.all(T.numpy() == 0)
    # the source goes away
    s = ek.sin(ek.Float32(x))
    assert bits_equal(ek.hsum(s).numpy(), eager["hsum_sin"]) and bits_equal(s.numpy(), eager["sin"])
    # sincos: one kernel fills both halves whichever is touched first
    l0 = ek.hip_launch_count()
    s, c = ek.sincos(dx)
    assert ek.hip_launch_count() - l0 == 0
    assert bits_equal(c.numpy(), eager["cos"])
    assert ek.hip_launch_count() - l0 == 1
    assert bits_equal(s.numpy(), eager["sin"])
    assert ek.hip_launch_count() - l0 == 1
    # one half reduced on load, the other evaluated on its own
    s, c = ek.sincos(dx)
    assert bits_equal(ek.hsum(s).numpy(), eager["hsum_sin"])
    assert bits_equal(c.numpy(), eager["cos"]) and bits_equal(s.numpy(), eager["sin"])
    # one half dropped before the other is evaluated
    s, c = ek.sincos(dx)
    del s
    assert bits_equal(c.numpy(), eager["cos"])
    # small arrays are evaluated right away (unless the suite runs with the threshold overridden)
    if not os.environ.get("ENOKI_HIP_DEFER_MIN"):
        l0 = ek.hip_launch_count()
        t = ek.sin(ek.Float32(x[:1000]))
        assert ek.hip_launch_count() - l0 >= 1


def test_deferred_map_in_backward(ek):
    """y = hsum(sin(A[i] * x + B[i])); backward(): cos(u) is applied inside the adjoint scatter_add -- same gradients as with
    deferred evaluation switched off (both sides accumulate unordered: per-bin bound)"""
    import enoki_amd.hip_autodiff as ad
    ad.hip_init(0)
    rng = np.random.default_rng(17)
    n, K = 1 << 20, 1 << 17
    A = rng.standard_normal(K).astype(np.float32); B = rng.standard_normal(K).astype(np.float32)
    x = rng.standard_normal(n).astype(np.float32); idx = rng.integers(0, K, n).astype(np.uint32)

    def run():
        dA, dB = ad.Float32(A), ad.Float32(B)
        ad.set_requires_gradient(dA); ad.set_requires_gradient(dB)
        di = ad.UInt32(idx)
        l0 = ad.hip_launch_count()
        y = ad.hsum(ad.sin(ad.fmadd(ad.gather(dA, di), ad.Float32(x), ad.gather(dB, di))))
        ad.backward(y)
        launches = ad.hip_launch_count() - l0
        return ad.detach(y).numpy(), ad.gradient(dA).numpy(), ad.gradient(dB).numpy(), launches

    y1, ga1, gb1, l1 = run()
    ad.hip_set_defer_gather(False)
    try:
        y0, ga0, gb0, l0 = run()
    finally:
        ad.hip_set_defer_gather(True)
    assert l1 < l0
    from conftest import cfg3b_truth
    t = cfg3b_truth(A, B, x, idx)          # |A|, |B|, |x| are not bounded by 1 here: the f32 terms are within 4 ulp(|u|) of exact
    scale = max(1.0, float(np.abs(A).max()) * float(np.abs(x).max()) + float(np.abs(B).max()))
    assert abs(float(y1[0]) - t["y"]) <= t["y_bound"] * scale and abs(float(y1[0]) - t["y"]) <= t["y_stat_bound"] * scale
    for g1, g0, name in ((ga1, ga0, "gA"), (gb1, gb0, "gB")):
        bound = t[name + "_bound"] * scale          # cnt * sum|terms| * 2^-24 per bin
        assert np.all(np.abs(g1 - t[name]) <= bound) and np.all(np.abs(g0 - t[name]) <= bound), name


def test_tape_suites_with_everything_deferred():
    """The tape programs of the parity suites are small (below the 64 Ki / 4096-element thresholds), so by default they never
    meet a deferred node.  ENOKI_HIP_DEFER_MIN=1 defers EVERY fusable unary result and every gather: the bit-exact tape
    parity suite and the reference's own autodiff tests must still pass."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ENOKI_HIP_DEFER_MIN="1")
    files = [os.path.join(root, "tests", f) for f in ("test_tape_parity.py", "test_reference_autodiff_gpu.py",
                                                      "test_reference_sources_gpu.py", "test_call_gpu.py")]
    out = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x"] + files, env=env, capture_output=True, text=True,
                         timeout=1500, cwd=root)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("exe_name", ["fuzz_tape_hip.bin", "fuzz_tape_hip_f64.bin"])
def test_fuzzed_tape_programs_deferred_equals_eager_on_the_device(exe_name):
    """tests/cpp/asan_tape.cpp (400 random differentiable programs: shared-index gathers, struct gathers through records,
    unary maps, select, mid-graph reductions, sincos) linked against the REAL library: values and gradients with every
    gather / unary result deferred are bit-identical to eager evaluation (deterministic scatter_add order)"""
    import os, subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", exe_name)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "fuzz_tape_hip: 400 fuzzed" in out.stdout
