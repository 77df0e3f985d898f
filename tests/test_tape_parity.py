"""Tape / DiffArray parity.

CPU (not gpu):  product Tape<T> + DiffArray<T> instantiated over the oracle array type (tests/cpp/tape_host.cpp)
                vs the UNMODIFIED reference build (oracle/_ref) -- bit-for-bit on every program except the ones whose
                edge weights go through rcp/rsqrt (class C: the AVX2 reference uses rcpps + Newton).  This pins the
                host logic of the tape: graph bookkeeping, sweep order, refcounting, specials, simplification.
GPU  (gpu):     product Tape over HIPArray<float> (tests/cpp/tape_hip.cpp -> libenoki-hip-autodiff.so -> C ABI)
                vs the reference build when it travelled with the tree, else vs the committed fixtures in
                tests/golden/tape_*.npz (generated from the reference build by tests/golden/make_golden.py).
"""
import os

import numpy as np
import pytest

import tape_lib as tl
from conftest import bits_equal

HAVE_REF = os.path.exists(os.path.join(tl.ROOT, "oracle", "_ref", "libenoki_ref.so"))
GOLDEN = os.path.join(tl.HERE, "golden")
NAMES = sorted(tl.suite().keys())


def reference_result(name, prog):
    """reference outputs for a suite program: live from oracle/_ref when present, else the golden fixture"""
    if HAVE_REF:
        return tl.run(tl.ref_fn(), prog)
    z = np.load(os.path.join(GOLDEN, f"tape_{name}.npz"))
    grads = [z[f"g{i}"] if f"g{i}" in z else None for i in range(int(z["n_grads"]))]
    return z["value"], grads


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref was not built (needs /root/reference)")
@pytest.mark.parametrize("name", NAMES)
def test_host_tape_matches_reference_build(name):
    prog = tl.suite()[name]
    rv, rg = tl.run(tl.ref_fn(), prog)
    hv, hg = tl.run(tl.host_lib().host_tape_program, prog)
    assert tl.host_lib().host_tape_live_nodes() == 0, "tape leaked nodes"
    if name in tl.TOLERANT or name == "sqrt_log":
        assert bits_equal(rv, hv) if name not in tl.CLASS_C_VALUES else np.allclose(rv, hv, rtol=3e-6, atol=3e-6)
        for a, b in zip(rg, hg):
            if a is not None:
                assert np.allclose(a, b, rtol=3e-6, atol=3e-6)
    else:
        assert bits_equal(rv, hv), name
        for a, b in zip(rg, hg):
            assert (a is None and b is None) or bits_equal(a, b), name


@pytest.mark.parametrize("name", NAMES)
def test_golden_fixtures_match_host_tape(name):
    """the committed fixtures (made from the reference build) agree with the product tape on the oracle arrays"""
    prog = tl.suite()[name]
    z = np.load(os.path.join(GOLDEN, f"tape_{name}.npz"))
    hv, hg = tl.run(tl.host_lib().host_tape_program, prog)
    exact = not (name in tl.TOLERANT or name == "sqrt_log")
    assert bits_equal(z["value"], hv) if name not in tl.CLASS_C_VALUES else np.allclose(z["value"], hv, rtol=3e-6, atol=3e-6)
    for i, g in enumerate(hg):
        if g is None:
            continue
        assert bits_equal(z[f"g{i}"], g) if exact else np.allclose(z[f"g{i}"], g, rtol=3e-6, atol=3e-6)


CLASS_C_PROGRAMS = ["div_rcp_rsqrt", "sw_trig", "sw_hyp", "sw_sum", "sw_cbrt_pow"]


def _band(name, what, got, scalar_row, avx2_row):
    """`got` is no further (in ulp, over the finite entries) from either row of the reference than the rows are from each
    other.  (Composite outputs cancel -- sums of several function values -- so the distances can be large in ulp for BOTH
    rows; the band is what is meaningful, see conftest.py on class C.)"""
    from conftest import ulp_diff
    g, s, a = (np.asarray(v, np.float32).ravel() for v in (got, scalar_row, avx2_row))
    ok = np.isfinite(g) & np.isfinite(s) & np.isfinite(a) & (np.abs(g) > 1e-30) & (np.abs(s) > 1e-30) & (np.abs(a) > 1e-30)
    ds, da, dr = ulp_diff(g[ok], s[ok]).max(), ulp_diff(g[ok], a[ok]).max(), ulp_diff(s[ok], a[ok]).max()
    assert ds <= dr and da <= dr, (name, what, int(ds), int(da), int(dr))


def _class_c_program(name, run_fn, value_is_order_dependent):
    prog = tl.suite()[name]
    zs = np.load(os.path.join(GOLDEN, f"tape_scalar_{name}.npz"))      # the reference's scalar row (oracle/Makefile refscalar)
    za = np.load(os.path.join(GOLDEN, f"tape_{name}.npz"))             # its AVX2 row
    v, g = tl.run(run_fn, prog)
    if not value_is_order_dependent:
        _band(name, "value", v, zs["value"], za["value"])
    for i, gi in enumerate(g):
        if gi is not None:
            _band(name, f"g{i}", gi, zs[f"g{i}"], za[f"g{i}"])
    if name == "div_rcp_rsqrt":
        # weights made of division, rcp and rsqrt only: the scalar row's bits
        assert bits_equal(g[0], zs["g0"]) and bits_equal(g[1], zs["g1"])


@pytest.mark.parametrize("name", CLASS_C_PROGRAMS)
def test_host_tape_class_c_programs_between_the_reference_rows(name):
    """the five tape programs whose values / edge weights go through rcp() or rsqrt(), against BOTH rows of the reference"""
    _class_c_program(name, tl.host_lib().host_tape_program, value_is_order_dependent=False)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CLASS_C_PROGRAMS)
def test_hip_tape_class_c_programs_between_the_reference_rows(name):
    _class_c_program(name, tl.hip_lib().hip_tape_program, value_is_order_dependent=name in tl.ORDER_DEPENDENT_ON_GPU)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_tape_matches_reference(name):
    prog = tl.suite()[name]
    rv, rg = reference_result(name, prog)
    import gc
    gc.collect()
    live_before = tl.hip_lib().hip_tape_live_nodes()     # the tape is process-global (other tests may hold arrays)
    gv, gg = tl.run(tl.hip_lib().hip_tape_program, prog)
    assert tl.hip_lib().hip_tape_live_nodes() == live_before, "tape leaked nodes"
    assert gv.shape == rv.shape
    if name in tl.TOLERANT and name not in tl.ORDER_DEPENDENT_ON_GPU:
        # rcp() inside the primal / the weights: a few ulp from the AVX2 reference, but bit-exact against the
        # product tape over the CPU oracle (same algorithm, exact division) when that library travelled
        assert np.allclose(rv, gv, rtol=3e-6, atol=3e-6), name
        for a, b in zip(rg, gg):
            assert (a is None and b is None) or np.allclose(a, b, rtol=3e-6, atol=3e-6), name
        if os.path.exists(os.path.join(tl.HERE, "cpp", "libtape_host.so")):
            hv, hg = tl.run(tl.host_lib().host_tape_program, prog)
            assert bits_equal(hv, gv), name
            for a, b in zip(hg, gg):
                assert (a is None and b is None) or bits_equal(a, b), name
        return
    if name not in tl.ORDER_DEPENDENT_ON_GPU:
        # purely vertical programs: bit-exact against the reference
        assert bits_equal(rv, gv), name
        for a, b in zip(rg, gg):
            assert (a is None and b is None) or bits_equal(a, b), name
        return
    # programs containing horizontal reductions / fp scatter_add: the primal reduction and every gradient that passed through
    # a reduction depend on the summation order (class D).  Two orders of the same n <= 1000 terms differ by a few ulp AT THE
    # SCALE of what is summed: the output's own magnitude when the terms have a sign, sqrt(n) entries of input magnitude when
    # they cancel.  16 of those ulps; the errors observed on the MI355X (tools/probe_tape_errors.py) are 0 .. 4.6 of them.
    n = max(a.size for a, _ in prog.inputs)
    unit = 2.0 ** -24 * np.sqrt(n) * max(float(np.abs(a).max()) for a, _ in prog.inputs)

    def tol(ref):
        return 16 * (2.0 ** -24 * float(np.abs(ref).max()) + unit)
    assert np.all(np.abs(gv - rv) <= tol(rv)), (name, float(np.abs(gv - rv).max()), tol(rv))
    for a, b in zip(rg, gg):
        if a is None:
            continue
        assert np.all(np.abs(a - b) <= tol(a)), (name, float(np.abs(a - b).max()), tol(a))


@pytest.mark.gpu
def test_hip_cfg3a_gradients_bit_exact_elementwise():
    """cfg3a: the gradients are purely elementwise given the seed -> bit-exact even though y = hsum(..) is not"""
    prog = tl.suite(n=100003)["cfg3a"]
    rv, rg = reference_result("cfg3a_big", prog) if HAVE_REF else (None, None)
    if not HAVE_REF:
        pytest.skip("needs the reference build for n = 100003")
    gv, gg = tl.run(tl.hip_lib().hip_tape_program, prog)
    assert bits_equal(rg[0], gg[0]) and bits_equal(rg[2], gg[2])
    from conftest import hsum_bound, hsum_depth, stat_sum_bound
    (a, _), (x, _), (b, _) = prog.inputs
    s64 = np.sin(a.astype(np.float64) * x + b)
    assert abs(float(gv[0]) - s64.sum()) <= hsum_bound(s64)                                   # worst case of our order
    assert abs(float(gv[0]) - s64.sum()) <= stat_sum_bound(s64, hsum_depth(s64.size))         # 5 sigma


SEEDS = list(range(24))


def same_grads(ref, got, prog):
    """bit-exact for vector inputs; the scalar leaf's gradient is a horizontal sum (class D: order dependent)"""
    for (arr, _), a, b in zip(prog.inputs, ref, got):
        if a is None and b is None:
            continue
        if a is None or b is None:
            return False
        if prog.mode == "backward" and arr.size == 1:
            if not np.allclose(a, b, rtol=1e-4, atol=1e-4):
                return False
        elif not bits_equal(a, b):
            return False
    return True


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref was not built (needs /root/reference)")
@pytest.mark.parametrize("seed", SEEDS)
def test_random_vertical_programs_host_tape_vs_reference(seed):
    for mode in ("backward", "forward"):
        prog = tl.random_program(seed, mode=mode)
        rv, rg = tl.run(tl.ref_fn(), prog)
        hv, hg = tl.run(tl.host_lib().host_tape_program, prog)
        assert bits_equal(rv, hv), (seed, mode)
        assert same_grads(rg, hg, prog) if mode == "backward" else bits_equal(rg[0], hg[0]), (seed, mode)
    assert tl.host_lib().host_tape_live_nodes() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEEDS)
def test_random_vertical_programs_hip_tape_bit_exact(seed):
    """GPU tape vs the product tape on the CPU oracle arrays (itself bit-exact vs the reference build, test above) and,
    when the reference build travelled with the tree, vs the reference directly"""
    for mode in ("backward", "forward"):
        prog = tl.random_program(seed, n=4099, mode=mode)
        gv, gg = tl.run(tl.hip_lib().hip_tape_program, prog)
        hv, hg = tl.run(tl.host_lib().host_tape_program, prog)
        assert bits_equal(hv, gv), (seed, mode)
        assert same_grads(hg, gg, prog) if mode == "backward" else bits_equal(hg[0], gg[0]), (seed, mode)
        if HAVE_REF:
            rv, rg = tl.run(tl.ref_fn(), prog)
            assert bits_equal(rv, gv)
            assert same_grads(rg, gg, prog) if mode == "backward" else bits_equal(rg[0], gg[0]), (seed, mode)


# ---- gather adjoints: batched / fused scatter_adds ------------------------------------------------------------------
GATHER_NAMES = sorted(tl.gather_suite().keys())


def _all_equal(ref, got):
    rv, rg = ref; gv, gg = got
    if not bits_equal(rv, gv):
        return False
    return all((a is None and b is None) or (a is not None and b is not None and bits_equal(a, b)) for a, b in zip(rg, gg))


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref was not built (needs /root/reference)")
@pytest.mark.parametrize("name", GATHER_NAMES)
def test_gather_adjoints_host_tape_vs_reference(name):
    prog = tl.gather_suite()[name]
    assert _all_equal(tl.run(tl.ref_fn(), prog), tl.run(tl.host_lib().host_tape_program, prog)), name
    assert tl.host_lib().host_tape_live_nodes() == 0


@pytest.mark.parametrize("name", GATHER_NAMES)
def test_gather_adjoints_golden_vs_host_tape(name):
    prog = tl.gather_suite()[name]
    z = np.load(os.path.join(GOLDEN, f"tape_gather_{name}.npz"))
    hv, hg = tl.run(tl.host_lib().host_tape_program, prog)
    assert bits_equal(z["value"], hv)
    for i, g in enumerate(hg):
        assert g is None or bits_equal(z[f"g{i}"], g), (name, i)


@pytest.mark.gpu
@pytest.mark.parametrize("name", GATHER_NAMES)
@pytest.mark.parametrize("n,k", [(1000, 37), ((1 << 19) + 3, 70001), ((1 << 18) + 1, 1 << 20)])
def test_gather_adjoints_hip_tape_bit_exact(name, n, k):
    """small: atomic path per stream; large: the fused multi-table binned path (exact integer-valued data, so the
    accumulation order does not matter and the comparison with the CPU tape is bit for bit)"""
    prog = tl.gather_suite(n=n, k=k)[name]
    import gc
    gc.collect()
    live_before = tl.hip_lib().hip_tape_live_nodes()
    got = tl.run(tl.hip_lib().hip_tape_program, prog)
    assert tl.hip_lib().hip_tape_live_nodes() == live_before, "tape leaked nodes"
    if n == 1000:
        z = np.load(os.path.join(GOLDEN, f"tape_gather_{name}.npz"))
        assert bits_equal(z["value"], got[0])
        for i, g in enumerate(got[1]):
            assert g is None or bits_equal(z[f"g{i}"], g), (name, i)
    assert _all_equal(tl.run(tl.host_lib().host_tape_program, prog), got), name


GATHER_SEEDS = list(range(32))


def _values_equal(ref, got):
    """exact equality of every value; the sign of a zero may differ: sweeps over arrays with host-known unit weights hand
    gradient buffers on unchanged (-0 stays -0) where the literal safe_mul(1, g) writes +0"""
    rv, rg = ref; gv, gg = got
    if not np.array_equal(rv, gv):
        return False
    return all((a is None and b is None) or (a is not None and b is not None and np.array_equal(a, b)) for a, b in zip(rg, gg))


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref was not built (needs /root/reference)")
@pytest.mark.parametrize("seed", GATHER_SEEDS)
def test_random_gather_programs_host_tape_vs_reference(seed):
    prog = tl.random_gather_program(seed)
    assert _all_equal(tl.run(tl.ref_fn(), prog), tl.run(tl.host_lib().host_tape_program, prog)), seed
    assert tl.host_lib().host_tape_live_nodes() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("seed", GATHER_SEEDS)
def test_random_gather_programs_hip_tape_bit_exact(seed):
    """GPU tape (deferred / batched / weight-fused gather adjoints) vs the product tape over the CPU oracle arrays (itself
    bit-exact vs the reference build, test above): small inputs take the atomic path, large ones the binned multi path"""
    for n, k in ((1000, 37), ((1 << 18) + 5, 70001)):
        prog = tl.random_gather_program(seed, n=n, k=k)
        import gc
        gc.collect()
        live_before = tl.hip_lib().hip_tape_live_nodes()
        got = tl.run(tl.hip_lib().hip_tape_program, prog)
        assert tl.hip_lib().hip_tape_live_nodes() == live_before, "tape leaked nodes"
        assert _values_equal(tl.run(tl.host_lib().host_tape_program, prog), got), (seed, n)
        if HAVE_REF and n == 1000:
            assert _values_equal(tl.run(tl.ref_fn(), prog), got), (seed, n)


# ---- two scatters into one buffer (reference tests/autodiff.cpp test30_scatter) ------------------------------------------
SCATTER_TWICE = sorted(tl.scatter_twice_suite().keys())


@pytest.mark.skipif(not tl.ref512_available(), reason="oracle/_ref/libenoki_ref512.so missing or no AVX-512 host")
@pytest.mark.parametrize("name", SCATTER_TWICE)
def test_scatter_twice_host_tape_vs_avx512_reference(name):
    prog = tl.scatter_twice_suite()[name]
    assert _all_equal(tl.run(tl.ref512_fn(), prog), tl.run(tl.host_lib().host_tape_program, prog)), name


@pytest.mark.parametrize("name", SCATTER_TWICE)
def test_scatter_twice_golden_vs_host_tape(name):
    prog = tl.scatter_twice_suite()[name]
    z = np.load(os.path.join(GOLDEN, f"tape_scatter_twice_{name}.npz"))
    hv, hg = tl.run(tl.host_lib().host_tape_program, prog)
    assert bits_equal(z["value"], hv)
    for i, g in enumerate(hg):
        assert g is None or bits_equal(z[f"g{i}"], g), (name, i)


@pytest.mark.gpu
@pytest.mark.parametrize("name", SCATTER_TWICE)
def test_scatter_twice_hip_tape_bit_exact(name):
    """the GPU tape on the pattern that crashes the pinned AVX2 reference: against the committed vectors (made from the
    AVX-512 reference build), the host tape and -- where the host can run it -- the AVX-512 reference build itself"""
    prog = tl.scatter_twice_suite()[name]
    got = tl.run(tl.hip_lib().hip_tape_program, prog)
    z = np.load(os.path.join(GOLDEN, f"tape_scatter_twice_{name}.npz"))
    assert bits_equal(z["value"], got[0])
    for i, g in enumerate(got[1]):
        assert g is None or bits_equal(z[f"g{i}"], g), (name, i)
    assert _all_equal(tl.run(tl.host_lib().host_tape_program, prog), got), name
    if tl.ref512_available():
        assert _all_equal(tl.run(tl.ref512_fn(), prog), got), name
