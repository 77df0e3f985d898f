"""The adjoint sums of the forward + adjoint kernel in 64-bit FIXED POINT (EK_BUCKETED_HINT_ADJOINT | EK_BUCKETED_HINT_BOUNDED,
csrc/bucketed_early.hip, round 6) through the C ABI.

What is checked
  * the same multisets as the lock protocol: y and both gradient tables inside the class-D bounds against float64 (the yardstick
    of tests/test_bucketed_gpu.py), for sin / cos in both roles, fma and product-then-sum ops, masks, ragged sizes;
  * what fixed point promises on top: a term is carried exactly from 2^-13 of its bound upwards, so every table entry sits within
    (count + 1) * 2^-37 * bound + half an ulp of the exact sum -- far inside class D -- and the gradient tables are BIT-IDENTICAL
    from run to run (the reference's scatter_add is fixed-order, dynamic.h:517-534; the lock protocol's sums depend on who got
    a lock first);
  * the cases fixed point cannot carry fall back to the locks per piece or per launch with the reference's semantics: an infinite
    or NaN x (no scale exists), a NaN table entry (a NaN term);
  * the hint is ignored where it does not apply (tables beyond 256 quarter-size buckets, ENOKI_HIP_EARLY_SUMS=locks).
"""
import numpy as np
import pytest

from conftest import uniform_pm1

pytestmark = pytest.mark.gpu
EPS = 2.0 ** -24


def up(capi, a):
    return capi.Buf.from_numpy(a)


def make(n, K, seed, xscale=1.0):
    rng = np.random.default_rng(seed)
    A = uniform_pm1(K, seed + 1).astype(np.float32)
    C = uniform_pm1(K, seed + 2).astype(np.float32)
    x = (uniform_pm1(n, seed + 3) * xscale).astype(np.float32)
    idx = rng.integers(0, K, n).astype(np.uint32)
    return A, C, x, idx


def launches(capi, fn):
    capi.profile_begin()
    fn()
    return {k["kernel"]: k["launches"] for k in capi.profile_end() if k["launches"]}


def truth(capi, op, half, dA, dC, dx, di, x, mask=None):
    """the EXACT multiset of f32 terms: u from the library's element-order kernel (class A: bit-identical to the reference build,
    tests/test_gathered_gpu.py), sin / cos of it from the library's own maps, the weighted term as the f32 product safe_mul forms.
    What remains between these and the kernel's tables is summation -- nothing else."""
    u = capi.map_gathered(op, capi.G(dA, di), dx, capi.G(dC, di))
    if mask is not None:
        un = u.numpy().copy()
        un[~mask] = 0.0                                   # masked-out lanes gather 0 from both tables: u = fma(0, x, 0)
        u = capi.Buf.from_numpy(un)
    other = "cos" if half == "sin" else "sin"
    red = capi.unary(half, u).numpy().astype(np.float64)
    kept32 = capi.unary(other, u).numpy()
    weighted32 = (kept32 * x).astype(np.float32)
    return red, kept32.astype(np.float64), weighted32.astype(np.float64)


def scale_ulp(n, bound):
    """what one unit of the fixed-point sums is worth: 2^-S with S = 62 - ceil(log2(n + 1)) - e, bound < 2^e (csrc/ek_bucketed.h)"""
    lg = 1
    while (1 << lg) <= n:
        lg += 1
    e = 0 if bound <= 0 else int(np.floor(np.log2(bound))) + 1
    return 2.0 ** -(62 - lg - e)


def run(capi, op, half, dA, dx, dC, di, K, hints, mask=None):
    other = "cos" if half == "sin" else "sin"
    b = capi.Bucketed(op, dA, dx, dC, di, hints=hints, mask=mask)
    y = float(b.reduce("hsum", half, keep=True, keep_op=other).numpy()[0])
    g0, g1 = up(capi, np.zeros(K, np.float32)), up(capi, np.zeros(K, np.float32))
    ks = launches(capi, lambda: b.scatter_add([g0, g1], [(other, 0, False), (other, 0, True)], fresh=[1, 1]))
    out = (y, g0.numpy().copy(), g1.numpy().copy(), ks)
    b.destroy()
    return out


@pytest.mark.parametrize("op", ["fmadd", "muladd"])
@pytest.mark.parametrize("half", ["sin", "cos"])
@pytest.mark.parametrize("masked", [False, True])
def test_fixed_point_sums_are_inside_class_d_and_reproducible(capi, op, half, masked):
    K, n = (1 << 18) + 5, (1 << 21) + 1237
    A, C, x, idx = make(n, K, seed=31, xscale=3.0)
    mask = (np.random.default_rng(5).integers(0, 4, n) != 0) if masked else None
    dA, dC, dx, di = up(capi, A), up(capi, C), up(capi, x), up(capi, idx)
    dm = up(capi, mask.astype(np.uint8)) if masked else None
    H = capi.Bucketed.HINT_ADJOINT | capi.Bucketed.HINT_BOUNDED
    b = capi.Bucketed(op, dA, dx, dC, di, hints=H, mask=dm)
    other = "cos" if half == "sin" else "sin"
    ks = launches(capi, lambda: b.reduce("hsum", half, keep=True, keep_op=other))
    assert ks.get("bucket_pair_fma_reduce_adjoint") == 1, ks
    b.destroy()
    y, g0, g1, ks = run(capi, op, half, dA, dx, dC, di, K, H, dm)
    assert ks == {"scatter_add_fold": 1}, ks
    red, kept, weighted = truth(capi, op, half, dA, dC, dx, di, x, mask)
    on = np.ones(n, bool) if mask is None else mask
    ii = idx.astype(np.int64)[on]
    cnt = np.bincount(ii, minlength=K)
    assert abs(y - red.sum()) <= EPS * (n // 4096 + 64) * np.abs(red).sum()        # (masked-out lanes: u = 0, they enter as f(0))
    xmax = float(np.abs(x).max())
    for g, terms, bound1 in ((g0, kept[on], 1.0), (g1, weighted[on], xmax)):
        ref = np.bincount(ii, weights=terms, minlength=K)
        sabs = np.bincount(ii, weights=np.abs(terms), minlength=K)
        err = np.abs(g.astype(np.float64) - ref)
        # class D: what every scatter_add of the library promises ...
        assert (err <= EPS * (cnt + 1) * sabs + 1e-300).all(), float((err / (EPS * (cnt + 1) * sabs + 1e-300)).max())
        # ... and what fixed point gives: less than one unit of the scale per term (rounded down), then ONE rounding of the sum
        assert (err <= cnt * scale_ulp(n, bound1) + EPS * np.abs(ref) + 1e-300).all()
    # bit-identical from run to run (three more runs, each with its own partition)
    for _ in range(3):
        y2, h0, h1, _ = run(capi, op, half, dA, dx, dC, di, K, H, dm)
        assert np.array_equal(g0.view(np.uint32), h0.view(np.uint32)) and np.array_equal(g1.view(np.uint32), h1.view(np.uint32))
        # ... and so is the reduced value: its terms are summed as integers (units of 2^-28), exactly
        assert np.float32(y2).view(np.uint32) == np.float32(y).view(np.uint32), (y, y2)
    # within n * 2^-28 (truncation of each term) + one rounding of the exact sum
    assert abs(y - red.sum()) <= n * 2.0 ** -28 + 2.0 ** -24 * abs(red.sum()) + 1e-6


def test_fixed_point_against_the_lock_protocol_on_the_same_input(capi):
    """both forms add the same multisets: they differ by roundings only, entry by entry within twice the class-D bound"""
    K, n = 1 << 20, 1 << 22
    A, C, x, idx = make(n, K, seed=77)
    dA, dC, dx, di = up(capi, A), up(capi, C), up(capi, x), up(capi, idx)
    yl, l0, l1, _ = run(capi, "fmadd", "sin", dA, dx, dC, di, K, capi.Bucketed.HINT_ADJOINT)
    yf, f0, f1, _ = run(capi, "fmadd", "sin", dA, dx, dC, di, K, capi.Bucketed.HINT_ADJOINT | capi.Bucketed.HINT_BOUNDED)
    cnt = np.bincount(idx.astype(np.int64), minlength=K)
    assert abs(yl - yf) <= EPS * 2048 * n
    for l, f in ((l0, f0), (l1, f1)):
        assert (np.abs(l.astype(np.float64) - f.astype(np.float64)) <= 2 * EPS * (cnt + 1) * (cnt + 1)).all()


@pytest.mark.parametrize("special", [np.inf, -np.inf, np.nan])
def test_non_finite_x_takes_the_locks_for_the_whole_launch(capi, special):
    """no scale exists: every piece runs under the exchange locks, results as without the BOUNDED hint (NaNs in the same entries)"""
    K, n = 1 << 18, (1 << 20) + 3
    A, C, x, idx = make(n, K, seed=9)
    x[12345] = special
    dA, dC, dx, di = up(capi, A), up(capi, C), up(capi, x), up(capi, idx)
    yl, l0, l1, _ = run(capi, "fmadd", "sin", dA, dx, dC, di, K, capi.Bucketed.HINT_ADJOINT)
    yf, f0, f1, ks = run(capi, "fmadd", "sin", dA, dx, dC, di, K, capi.Bucketed.HINT_ADJOINT | capi.Bucketed.HINT_BOUNDED)
    assert ks == {"scatter_add_fold": 1}
    assert np.isnan(yl) and np.isnan(yf)
    for l, f in ((l0, f0), (l1, f1)):
        assert np.array_equal(np.isnan(l), np.isnan(f)) and np.isnan(f[idx[12345]])
        ok = ~np.isnan(l)
        assert np.allclose(l[ok], f[ok], rtol=0, atol=1e-3)


def test_nan_table_entry_redoes_the_piece_under_locks(capi):
    """a NaN in A: the terms of that entry are NaN -- fixed point cannot carry them, the piece is redone under locks; the entry's
    gradients are NaN as in the reference (the sum of its terms), every other entry is untouched"""
    K, n = 1 << 18, (1 << 20) + 3
    A, C, x, idx = make(n, K, seed=10)
    bad = int(idx[777])
    A[bad] = np.nan
    dA, dC, dx, di = up(capi, A), up(capi, C), up(capi, x), up(capi, idx)
    H = capi.Bucketed.HINT_ADJOINT | capi.Bucketed.HINT_BOUNDED
    yf, f0, f1, _ = run(capi, "fmadd", "sin", dA, dx, dC, di, K, H)
    assert np.isnan(yf) and np.isnan(f0[bad]) and np.isnan(f1[bad])
    assert np.isnan(f0).sum() == 1 and np.isnan(f1).sum() == 1
    dA0 = up(capi, np.nan_to_num(A))
    red, kept, weighted = truth(capi, "fmadd", "sin", dA0, dC, dx, di, x)
    ii = idx.astype(np.int64)
    ref = np.bincount(ii, weights=kept, minlength=K)
    cnt = np.bincount(ii, minlength=K)
    ok = np.arange(K) != bad
    assert (np.abs(f0.astype(np.float64) - ref)[ok] <= EPS * (cnt + 1)[ok] * (cnt + 1)[ok]).all()


@pytest.mark.parametrize("xscale", [0.0, 1e-20, 1e-30, 1e-42, 64.0])
def test_extreme_scales_of_x(capi, xscale):
    """max |x| zero, small, tiny, denormal, huge: the scale of the weighted sums follows max |x|; below ~1e-26 (where 2^S would leave
    the normal range) the launch runs under locks; every entry within class D either way"""
    K, n = 1 << 18, (1 << 20) + 11
    A, C, x, idx = make(n, K, seed=12, xscale=xscale)
    dA, dC, dx, di = up(capi, A), up(capi, C), up(capi, x), up(capi, idx)
    H = capi.Bucketed.HINT_ADJOINT | capi.Bucketed.HINT_BOUNDED
    y, g0, g1, _ = run(capi, "fmadd", "cos", dA, dx, dC, di, K, H)
    red, kept, weighted = truth(capi, "fmadd", "cos", dA, dC, dx, di, x)
    ii = idx.astype(np.int64)
    cnt = np.bincount(ii, minlength=K)
    for g, terms in ((g0, kept), (g1, weighted)):
        ref = np.bincount(ii, weights=terms, minlength=K)
        sabs = np.bincount(ii, weights=np.abs(terms), minlength=K)
        err = np.abs(g.astype(np.float64) - ref)
        tiny = 2.0 ** -149
        assert (err <= EPS * (cnt + 1) * sabs + (cnt + 1) * tiny).all(), (xscale, float(err.max()))


def test_hint_is_ignored_beyond_256_quarter_size_buckets(capi):
    """K = 2 Mi: half-size buckets under locks as in round 5 (the BOUNDED hint asks for nothing the table cannot give)"""
    K, n = 1 << 21, 1 << 21
    A, C, x, idx = make(n, K, seed=3)
    dA, dC, dx, di = up(capi, A), up(capi, C), up(capi, x), up(capi, idx)
    H = capi.Bucketed.HINT_ADJOINT | capi.Bucketed.HINT_BOUNDED
    y, g0, g1, ks = run(capi, "fmadd", "sin", dA, dx, dC, di, K, H)
    red, kept, weighted = truth(capi, "fmadd", "sin", dA, dC, dx, di, x)
    ii = idx.astype(np.int64)
    cnt = np.bincount(ii, minlength=K)
    assert (np.abs(g0.astype(np.float64) - np.bincount(ii, weights=kept, minlength=K)) <= EPS * (cnt + 1) * (cnt + 1)).all()


def test_every_launch_reduces_its_own_partials(capi):
    """The finish ticket of the bucket kernels is a RELAXED agent-scope atomic behind a wait for the wave's stores (csrc/ek_bucketed.h:
    finish_ticket) -- no release fence.  A stale per-workgroup partial would be invisible wherever one step is repeated (yesterday's
    partial equals today's), so: new values on every launch, the reduced value against float64 every time, at a size at which all
    workgroups finish together (tools/stress_finish_ticket.py is the long form: 1200 launches, profiles/stress_finish_ticket_r06.txt)."""
    rng = np.random.default_rng(11)
    n, K = 1 << 20, 1 << 20
    idx_h = rng.integers(0, K, n).astype(np.uint32)
    di = up(capi, idx_h)
    for r in range(24):
        a_h, c_h = rng.uniform(-1, 1, K).astype(np.float32), rng.uniform(-1, 1, K).astype(np.float32)
        x_h = (rng.uniform(-1, 1, n) * (1 + r % 3)).astype(np.float32)
        dA, dC, dx = up(capi, a_h), up(capi, c_h), up(capi, x_h)
        u = a_h[idx_h].astype(np.float64) * x_h + c_h[idx_h]
        for half, hints in (("sin", capi.Bucketed.HINT_ADJOINT | capi.Bucketed.HINT_BOUNDED), ("exp", capi.Bucketed.HINT_ADJOINT)):
            b = capi.Bucketed("fmadd", dA, dx, dC, di, hints=hints)
            y = float(b.reduce("hsum", half, keep=True, keep_op="cos" if half == "sin" else "exp").numpy()[0])
            b.destroy()
            terms = np.sin(u) if half == "sin" else np.exp(u)
            # (a workgroup's share is 1 / 256 of the terms; class D is ~1e-7 of the sum of magnitudes)
            assert abs(y - float(terms.sum())) <= 2e-6 * float(np.abs(terms).sum()), (r, half, y, float(terms.sum()))
