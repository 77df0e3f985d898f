"""Bucket-ordered gather -> fma -> {reduction, scatter_add} (ek_hip_bucketed_*, csrc/bucketed.hip) through the C ABI, checked against
the library's OWN element-order kernels and float64 numpy -- not against oracle/ directly (the whole step is compared with the
reference build at 64 Mi elements in test_headline_parity_gpu.py, test_neighbours_gpu.py and inside bench.py).

Yardstick: the element-order kernels of the same library give u = fma(A[idx], x, C[idx]) bit for bit (class A, pinned against
the reference build in test_gathered_gpu.py / test_kernels_gpu.py); the bucket-ordered path must
  * reduce exactly the multiset { map(u_i) } -- checked against the float64 sum of the f32 terms with the class-D bound of
    OUR summation depth, and a statistical sqrt(n) bound ten times tighter than the worst case,
  * add exactly the multiset of contributions per bin -- float64 bincount of the f32 terms, bound cnt * sum|terms| * 2^-24,
    and EXACTLY (bit for bit) for integer-valued data, where the order of the additions cannot matter,
  * give hmin / hmax bit for bit (order independent).
"""
import numpy as np
import pytest

from conftest import bits_equal, uniform_pm1

pytestmark = pytest.mark.gpu

EPS = {np.float32: 2.0 ** -24, np.float64: 2.0 ** -53}


def up(capi, a):
    return capi.Buf.from_numpy(a)


def make(dtype, n, K, seed, integer=False):
    rng = np.random.default_rng(seed)
    if integer:
        A = rng.integers(-3, 4, K).astype(dtype); C = rng.integers(-3, 4, K).astype(dtype); x = rng.integers(-2, 3, n).astype(dtype)
    else:
        A = uniform_pm1(K, seed + 1).astype(dtype); C = uniform_pm1(K, seed + 2).astype(dtype); x = uniform_pm1(n, seed + 3).astype(dtype)
    idx = rng.integers(0, K, n).astype(np.uint32)
    return A, C, x, idx


def element_order_u(capi, op, dA, dx, dC, di):
    return capi.map_gathered(op, capi.G(dA, di), dx, capi.G(dC, di))


def depth(n):
    """additions behind one output of the bucket-ordered reduction: 4 accumulators per lane of a 1024-lane workgroup over a
    piece of at least 32 Ki elements (at most n / 256 for large inputs), then 2 + 6 + 4 tree levels, then <= 768 partials
    in a 256-lane workgroup (3 + 6 + 2)"""
    piece = max(32768, n // 256 + 1)
    return piece // 4096 + 1 + 12 + 11


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("op", ["fmadd", "fmsub", "fnmadd", "fnmsub"])
def test_reduce_matches_element_order(capi, dtype, op):
    K = (1 << 16) + 77 if dtype == np.float32 else (1 << 15) + 77       # 5 buckets, a ragged last one
    n = (1 << 19) + 12345
    A, C, x, idx = make(dtype, n, K, seed=11)
    assert capi.Bucketed.applicable(dtype, np.uint32, K, n)
    dA, dC, dx, di = up(capi, A), up(capi, C), up(capi, x), up(capi, idx)
    u = element_order_u(capi, op, dA, dx, dC, di).numpy()
    for rop, mop in (("hsum", "sin"), ("hsum", None), ("hmax", "cos"), ("hmin", None), ("hsum", "exp"), ("hprod", "cos")):
        b = capi.Bucketed(op, dA, dx, dC, di)
        got = b.reduce(rop, mop, keep=False).numpy()[0]
        terms = capi.unary(mop, up(capi, u)).numpy() if mop else u
        t64 = terms.astype(np.float64)
        if rop == "hsum":
            err = abs(float(got) - t64.sum())
            assert err <= EPS[dtype] * depth(n) * np.abs(t64).sum(), (rop, mop, err)
            # statistically: roundings are independent, so the sequential chains contribute like a random walk over the
            # terms (sqrt(depth * sum t^2)) and the tree levels a few ulps of the total
            assert err <= EPS[dtype] * (8 * np.sqrt(depth(n) * (t64 ** 2).sum()) + 4 * abs(t64.sum())), (rop, mop, err)
        elif rop == "hprod":
            ref = np.exp(np.log(np.abs(t64)).sum())
            assert abs(abs(float(got)) - ref) <= 4 * n * EPS[dtype] * ref + 1e-300
        else:
            ref = terms.max() if rop == "hmax" else terms.min()
            assert bits_equal(np.array([got]), np.array([ref])), (rop, mop)
        b.destroy()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_kept_values_are_the_element_order_values(capi, dtype):
    """reduce(keep) then reductions over the kept u: the same multiset as the element-order u (sorted values equal)"""
    K, n = (1 << 17) + 5, (1 << 20) + 3
    A, C, x, idx = make(dtype, n, K, seed=3)
    dA, dC, dx, di = up(capi, A), up(capi, C), up(capi, x), up(capi, idx)
    u = element_order_u(capi, "fmadd", dA, dx, dC, di).numpy()
    b = capi.Bucketed("fmadd", dA, dx, dC, di)
    b.reduce("hsum", "sin", keep=True)
    for rop in ("hmin", "hmax"):             # now served from the kept values
        got = b.reduce(rop, None).numpy()[0]
        assert bits_equal(np.array([got]), np.array([u.min() if rop == "hmin" else u.max()]))
    got = float(b.reduce("hsum", "abs").numpy()[0])
    assert abs(got - np.abs(u.astype(np.float64)).sum()) <= EPS[dtype] * depth(n) * np.abs(u.astype(np.float64)).sum()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_scatter_add_integer_data_is_exact(capi, dtype):
    """integer-valued tables and x: every product and partial sum is exact, so any order gives the same bits"""
    K = (1 << 16) + 9 if dtype == np.float32 else (1 << 15) + 9
    n = (1 << 19) + 777
    A, C, x, idx = make(dtype, n, K, seed=5, integer=True)
    dA, dC, dx, di = up(capi, A), up(capi, C), up(capi, x), up(capi, idx)
    u = A[idx] * x + C[idx]
    ii = idx.astype(np.int64)
    b = capi.Bucketed("fmadd", dA, dx, dC, di)
    # four streams (two launches): |u|, x * |u|, the constant 1 weighted by x (= x), the constant 2
    T = [up(capi, np.full(K, 1.0, dtype)) for _ in range(4)]
    b.scatter_add(T, [("abs", 0, False), ("abs", 0, True), (None, 1.0, True), (None, 2.0, False)])
    refs = [np.abs(u), x * np.abs(u), x, np.full(n, 2.0)]
    for t, r in zip(T, refs):
        want = 1.0 + np.bincount(ii, weights=r.astype(np.float64), minlength=K)
        assert np.array_equal(t.numpy().astype(np.float64), want)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_scatter_add_cos_pair_within_class_d(capi, dtype):
    """the adjoint of BASELINE config 3b: gA[idx] += x * cos(u), gC[idx] += cos(u), after a forward hsum(sin(u))"""
    K, n = (1 << 16) + 1, (1 << 20) + 17
    A, C, x, idx = make(dtype, n, K, seed=9)
    dA, dC, dx, di = up(capi, A), up(capi, C), up(capi, x), up(capi, idx)
    u = element_order_u(capi, "fmadd", dA, dx, dC, di)
    cu = capi.unary("cos", u).numpy().astype(np.float64)
    ii = idx.astype(np.int64)
    cnt = np.bincount(ii, minlength=K)
    for forward_first in (True, False, "partner"):
        b = capi.Bucketed("fmadd", dA, dx, dC, di)
        if forward_first == "partner":
            # the forward keeps cos(u) -- the other half of the sincos it evaluates -- instead of u
            y = float(b.reduce("hsum", "sin", keep=True, keep_op="cos").numpy()[0])
            s64 = capi.unary("sin", u).numpy().astype(np.float64)
            assert abs(y - s64.sum()) <= EPS[dtype] * depth(n) * np.abs(s64).sum()
            # a stream that wants another function of u afterwards: u is rebuilt in bucket order
            t = up(capi, np.zeros(K, dtype))
            b.scatter_add([t], [("sin", 0, False)])
            ref = np.bincount(ii, weights=s64, minlength=K)
            assert (np.abs(t.numpy().astype(np.float64) - ref) <= EPS[dtype] * cnt * np.bincount(ii, weights=np.abs(s64), minlength=K) + 1e-300).all()
            # ... and the kept half reduced directly
            c1 = float(b.reduce("hsum", "cos").numpy()[0])
            assert abs(c1 - cu.sum()) <= EPS[dtype] * (n // (1 << 20) + 40) * np.abs(cu).sum()
        elif forward_first:
            b.reduce("hsum", "sin", keep=True)
        gA, gC = up(capi, np.zeros(K, dtype)), up(capi, np.zeros(K, dtype))
        b.scatter_add([gC, gA], [("cos", 0, False), ("cos", 0, True)])
        for got, terms in ((gA, cu * x.astype(np.float64)), (gC, cu)):
            # the product x * cos(u) is rounded once more in the working precision
            tr = terms.astype(dtype).astype(np.float64)
            ref = np.bincount(ii, weights=tr, minlength=K)
            bound = EPS[dtype] * (cnt * np.bincount(ii, weights=np.abs(tr), minlength=K)) + 1e-300
            err = np.abs(got.numpy().astype(np.float64) - ref)
            assert (err <= bound).all(), (forward_first, float((err / bound).max()))


FUSABLE = ["neg", "abs", "sqrt", "rcp", "rsqrt", "sin", "cos", "exp", "log", "rcp_sqr", "rsqrt_sqr", "rsqrt_cube"]


@pytest.mark.parametrize("mop", FUSABLE)
def test_scatter_add_applies_every_fusable_op(capi, mop):
    """Every op that unary_fusable() accepts, as the value stream of a bucket-ordered scatter_add in each of the kernel's forms:
    the compile-time pair { f(u), x f(u) } (Spec), a uniform pair with both streams unweighted, one stream alone, and a pair of
    DIFFERENT ops (run-time form).  (Round 4 shipped a switch that sent rsqrt_sqr to the copy body: the streams scattered u
    instead of rsqrt(u)^2.)  u = A x + C with C in [2, 4): every op is finite and well away from its singularities."""
    dtype = np.float32
    K, n = (1 << 16) + 1, (1 << 19) + 17
    A, C, x, idx = make(dtype, n, K, seed=31)
    A = (A * np.float32(0.5)).astype(dtype); C = (C + np.float32(3.0)).astype(dtype)
    dA, dC, dx, di = up(capi, A), up(capi, C), up(capi, x), up(capi, idx)
    u = element_order_u(capi, "fmadd", dA, dx, dC, di)
    fu = capi.unary(mop, u).numpy().astype(np.float64)
    other = "cos" if mop != "cos" else "sin"
    gu = capi.unary(other, u).numpy().astype(np.float64)
    ii = idx.astype(np.int64)
    cnt = np.bincount(ii, minlength=K)
    x64 = x.astype(np.float64)

    def check(got, terms, what):
        tr = terms.astype(dtype).astype(np.float64)
        ref = np.bincount(ii, weights=tr, minlength=K)
        bound = EPS[dtype] * (cnt * np.bincount(ii, weights=np.abs(tr), minlength=K)) + 1e-300
        err = np.abs(got.numpy().astype(np.float64) - ref)
        assert (err <= bound).all(), (mop, what, float((err / bound).max()))

    b = capi.Bucketed("fmadd", dA, dx, dC, di)
    t0, t1 = up(capi, np.zeros(K, dtype)), up(capi, np.zeros(K, dtype))
    b.scatter_add([t0, t1], [(mop, 0, False), (mop, 0, True)])                    # Spec: { f(u), x f(u) }
    check(t0, fu, "spec plain"); check(t1, fu * x64, "spec weighted")
    t0, t1 = up(capi, np.zeros(K, dtype)), up(capi, np.zeros(K, dtype))
    b.scatter_add([t0, t1], [(mop, 0, False), (mop, 0, False)])                   # uniform, both unweighted
    check(t0, fu, "uniform 0"); check(t1, fu, "uniform 1")
    t0 = up(capi, np.zeros(K, dtype))
    b.scatter_add([t0], [(mop, 0, True)])                                          # one stream
    check(t0, fu * x64, "single weighted")
    t0, t1 = up(capi, np.zeros(K, dtype)), up(capi, np.zeros(K, dtype))
    b.scatter_add([t0, t1], [(mop, 0, False), (other, 0, True)])                  # two different ops: run-time form
    check(t0, fu, "mixed 0"); check(t1, gu * x64, "mixed 1")
    b.destroy()


def launches(capi, fn):
    capi.profile_begin()
    fn()
    out = {}
    for k in capi.profile_end():
        if k["launches"]:
            out[k["kernel"]] = out.get(k["kernel"], 0) + k["launches"]
    return out


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("op", ["fmadd", "fnmsub"])
@pytest.mark.parametrize("half", ["sin", "cos"])
def test_hinted_plan_forms_the_adjoint_in_the_forward_pass(capi, dtype, op, half):
    """EK_BUCKETED_HINT_ADJOINT: reduce(hsum, sin, keep cos) sums cos(u) and x cos(u) per table entry in the same pass (half-size
    buckets: table slice and gradient tables share the LDS); the scatter_add of exactly those streams only folds.  Same
    multisets as the unhinted object, so the same class-D bounds; fresh targets are written, others added to."""
    other = "cos" if half == "sin" else "sin"
    K, n = (1 << 16) + 1, (1 << 20) + 17
    A, C, x, idx = make(dtype, n, K, seed=21)
    dA, dC, dx, di = up(capi, A), up(capi, C), up(capi, x), up(capi, idx)
    u = element_order_u(capi, op, dA, dx, dC, di)
    red = capi.unary(half, u).numpy().astype(np.float64)
    kept = capi.unary(other, u).numpy().astype(np.float64)
    ii = idx.astype(np.int64)
    cnt = np.bincount(ii, minlength=K)
    b = capi.Bucketed(op, dA, dx, dC, di, hints=capi.Bucketed.HINT_ADJOINT)
    ks = launches(capi, lambda: b.reduce("hsum", half, keep=True, keep_op=other))
    assert ks.get("bucket_pair_fma_reduce_adjoint") == 1 and "bucket_pair_fma_reduce" not in ks, ks
    b2 = capi.Bucketed(op, dA, dx, dC, di, hints=capi.Bucketed.HINT_ADJOINT)
    y = float(b2.reduce("hsum", half, keep=True, keep_op=other).numpy()[0])
    assert abs(y - red.sum()) <= EPS[dtype] * depth(n) * np.abs(red).sum()
    for b_, order, fresh in ((b, (False, True), (1, 0)), (b2, (True, False), (0, 0))):
        # targets: one "fresh" (garbage that must be overwritten), one holding 1.0 (added to);  both orders of the two streams
        T = [up(capi, np.full(K, 7.0 if f else 1.0, dtype)) for f in fresh]
        ks = launches(capi, lambda: b_.scatter_add(T, [(other, 0, w) for w in order], fresh=list(fresh)))
        assert ks == {"scatter_add_fold": 1}, ks
        for t, w, f in zip(T, order, fresh):
            terms = (kept * x.astype(np.float64) if w else kept).astype(dtype).astype(np.float64)
            ref = (0.0 if f else 1.0) + np.bincount(ii, weights=terms, minlength=K)
            bound = EPS[dtype] * ((cnt + 1) * (1.0 + np.bincount(ii, weights=np.abs(terms), minlength=K))) + 1e-300
            err = np.abs(t.numpy().astype(np.float64) - ref)
            assert (err <= bound).all(), (w, f, float((err / bound).max()))
    # one stream only, and a second fold of the same sums
    t = up(capi, np.zeros(K, dtype))
    b.scatter_add([t], [(other, 0, True)])
    b.scatter_add([t], [(other, 0, True)])
    terms = (kept * x.astype(np.float64)).astype(dtype).astype(np.float64)
    ref = 2.0 * np.bincount(ii, weights=terms, minlength=K)
    assert (np.abs(t.numpy().astype(np.float64) - ref) <= 2 * EPS[dtype] * (cnt + 1) * np.bincount(ii, weights=np.abs(terms), minlength=K) + 1e-300).all()
    # anything else on the same object: the ordinary kernels (u is rebuilt in bucket order)
    t = up(capi, np.zeros(K, dtype))
    ks = launches(capi, lambda: b.scatter_add([t], [(half, 0, False)]))
    assert ks.get("bucket_accumulate") == 1, ks
    ref = np.bincount(ii, weights=red, minlength=K)
    assert (np.abs(t.numpy().astype(np.float64) - ref) <= EPS[dtype] * cnt * np.bincount(ii, weights=np.abs(red), minlength=K) + 1e-300).all()
    um = u.numpy()
    assert bits_equal(np.array([b.reduce("hmax", None).numpy()[0]]), np.array([um.max()]))
    b.destroy(); b2.destroy()


def test_hinted_plan_integer_data_is_exact_and_hint_can_be_ignored(capi):
    """A = C = 0: every u is 0, the kept half cos(0) is exactly 1, and the sums are exact counts / exact sums of the
    integer-valued x -- any order of additions gives the same bits"""
    K, n = (1 << 16) + 3, (1 << 19) + 5
    rng = np.random.default_rng(2)
    A = np.zeros(K, np.float32); x = rng.integers(-2, 3, n).astype(np.float32)
    idx = rng.integers(0, K, n).astype(np.uint32)
    dA, dx, di = up(capi, A), up(capi, x), up(capi, idx)
    for tuning in (1, 0):
        capi.set_tuning("early_adjoint", tuning)
        try:
            b = capi.Bucketed("fmadd", dA, dx, dA, di, hints=capi.Bucketed.HINT_ADJOINT)
            ks = launches(capi, lambda: b.reduce("hsum", "sin", keep=True, keep_op="cos"))
            assert ("bucket_pair_fma_reduce_adjoint" in ks) == bool(tuning), ks
            g1, gx = up(capi, np.zeros(K, np.float32)), up(capi, np.zeros(K, np.float32))
            b.scatter_add([g1, gx], [("cos", 0, False), ("cos", 0, True)])
            assert np.array_equal(g1.numpy(), np.bincount(idx, minlength=K).astype(np.float32))
            assert np.array_equal(gx.numpy(), np.bincount(idx, weights=x.astype(np.float64), minlength=K).astype(np.float32))
            b.destroy()
        finally:
            capi.set_tuning("early_adjoint", 1)
    # tables beyond 256 half-size buckets (round 4): cut into slices of 256 half-size buckets, one partition pass per slice,
    # the early adjoint in every slice
    K2 = (200 << 14) + 5
    A2 = up(capi, np.zeros(K2, np.float32))
    i2 = up(capi, rng.integers(0, K2, n).astype(np.uint32))
    b = capi.Bucketed("fmadd", A2, dx, A2, i2, hints=capi.Bucketed.HINT_ADJOINT)
    ks = launches(capi, lambda: b.reduce("hsum", "sin", keep=True, keep_op="cos"))
    assert ks.get("bucket_pair_fma_reduce_adjoint") == 2 and "bucket_pair_fma_reduce" not in ks, ks
    b.destroy()


def test_not_applicable_shapes_are_refused(capi):
    assert not capi.Bucketed.applicable(np.float32, np.uint32, 1 << 14, 1 << 20)        # one bucket
    assert capi.Bucketed.applicable(np.float32, np.uint32, (256 << 14) + 1, 1 << 20)      # more than 256 buckets: slices (f32)
    assert not capi.Bucketed.applicable(np.float64, np.uint32, (256 << 13) + 1, 1 << 20)  # more than 256 buckets
    assert not capi.Bucketed.applicable(np.float32, np.uint32, 1 << 20, 1 << 17)        # too few lookups
    assert not capi.Bucketed.applicable(np.int32, np.uint32, 1 << 20, 1 << 20)
    capi.set_tuning("deterministic", 1)
    try:
        assert not capi.Bucketed.applicable(np.float32, np.uint32, 1 << 20, 1 << 20)
        A = up(capi, np.zeros(1 << 20, np.float32)); x = up(capi, np.zeros(1 << 20, np.float32)); i = up(capi, np.zeros(1 << 20, np.uint32))
        with pytest.raises(capi.EnokiHipError):
            capi.Bucketed("fmadd", A, x, A, i)
    finally:
        capi.set_tuning("deterministic", 0)


@pytest.mark.parametrize("hinted", [False, True])
def test_tables_beyond_256_buckets_are_sliced(capi, hinted):
    """K = 5 Mi + 3 entries: 2 slices of 4 Mi unhinted / 3 slices of 2 Mi hinted; integer data -> exact sums, so the sliced object
    must agree EXACTLY with numpy, masked-out lanes included (their u = 0 enters the reduction as cos(0) = 1)"""
    K, n = (5 << 20) + 3, (1 << 20) + 77
    rng = np.random.default_rng(5)
    A = rng.integers(-3, 4, K).astype(np.float32); C = rng.integers(-3, 4, K).astype(np.float32)
    x = rng.integers(-2, 3, n).astype(np.float32)
    idx = rng.integers(0, K, n).astype(np.uint32)
    mask = rng.integers(0, 4, n) != 0
    dA, dC, dx, di, dm = up(capi, A), up(capi, C), up(capi, x), up(capi, idx), up(capi, mask.astype(np.uint8))
    b = capi.Bucketed("fmadd", dA, dx, dC, di, hints=capi.Bucketed.HINT_ADJOINT if hinted else 0, mask=dm)
    u = np.where(mask, A[idx] * x + C[idx], 0).astype(np.float32)
    # f = abs keeps integers: y = sum |u|; kept function for the adjoint: abs(u) as well ({f, f} pair)
    y = float(b.reduce("hsum", "abs", keep=True, keep_op="abs").numpy()[0])
    assert y == float(np.abs(u).astype(np.float64).sum())
    gA, gC = capi.fill(np.float32, 0, K), capi.fill(np.float32, 0, K)
    b.scatter_add([gC, gA], [("abs", 0, False), ("abs", 0, True)])
    eA = np.bincount(idx[mask], weights=(np.abs(u) * x)[mask], minlength=K).astype(np.float32)
    eC = np.bincount(idx[mask], weights=np.abs(u)[mask], minlength=K).astype(np.float32)
    assert np.array_equal(gA.numpy(), eA) and np.array_equal(gC.numpy(), eC)
    assert float(b.reduce("hmax", "neg", keep=False).numpy()[0]) == float((-u).max())
    b.destroy()


def test_slices_with_no_elements_and_out_of_range_indices(capi):
    """five hinted slices (K = 9 Mi: split by slice first), every index in slices 0 and 3: the other slices are empty -- their part of
    a fresh gradient table must still hold zeros -- and indices beyond the table are dropped like masked-out lanes"""
    K, n = 9 << 20, (1 << 19) + 11
    rng = np.random.default_rng(6)
    A = rng.integers(-3, 4, K).astype(np.float32); C = rng.integers(-3, 4, K).astype(np.float32)
    x = rng.integers(-2, 3, n).astype(np.float32)
    idx = np.where(rng.integers(0, 2, n) == 0, rng.integers(0, 2 << 20, n), rng.integers(6 << 20, 8 << 20, n)).astype(np.uint32)
    idx[::1000] = K + 5                                   # out of range
    ok = idx < K
    dA, dC, dx, di = up(capi, A), up(capi, C), up(capi, x), up(capi, idx)
    b = capi.Bucketed("fmadd", dA, dx, dC, di, hints=capi.Bucketed.HINT_ADJOINT)
    u = np.where(ok, A[np.minimum(idx, K - 1)] * x + C[np.minimum(idx, K - 1)], 0).astype(np.float32)
    y = float(b.reduce("hsum", "abs", keep=True, keep_op="abs").numpy()[0])
    assert y == float(np.abs(u).astype(np.float64).sum())
    gA, gC = capi.Buf(np.float32, K), capi.Buf(np.float32, K)          # uninitialised: fresh tables are WRITTEN
    b.scatter_add([gC, gA], [("abs", 0, False), ("abs", 0, True)], fresh=[1, 1])
    eA = np.bincount(idx[ok], weights=(np.abs(u) * x)[ok], minlength=K).astype(np.float32)
    eC = np.bincount(idx[ok], weights=np.abs(u)[ok], minlength=K).astype(np.float32)
    assert np.array_equal(gA.numpy(), eA) and np.array_equal(gC.numpy(), eC)
    b.destroy()


def test_skewed_indices(capi):
    """all lookups in one bucket / one bin: the exchange lock's wave-combining path and the piece split stay correct"""
    K, n = 1 << 18, (1 << 19) + 5
    rng = np.random.default_rng(1)
    A = rng.integers(-3, 4, K).astype(np.float32); C = rng.integers(-3, 4, K).astype(np.float32)
    x = rng.integers(-2, 3, n).astype(np.float32)
    for idx in (np.full(n, 70001, np.uint32), (rng.integers(0, 100, n) + 5 * 16384).astype(np.uint32),
                np.where(rng.integers(0, 2, n) == 0, 3, rng.integers(0, K, n)).astype(np.uint32)):
        dA, dC, dx, di = up(capi, A), up(capi, C), up(capi, x), up(capi, idx)
        u = A[idx] * x + C[idx]
        b = capi.Bucketed("fmadd", dA, dx, dC, di)
        got = float(b.reduce("hsum", None, keep=True).numpy()[0])
        assert got == float(u.astype(np.float64).sum())
        g = up(capi, np.zeros(K, np.float32))
        b.scatter_add([g], [("copy", 0, True)])
        want = np.bincount(idx.astype(np.int64), weights=(x * u).astype(np.float64), minlength=K)
        assert np.array_equal(g.numpy().astype(np.float64), want)


@pytest.mark.parametrize("hinted", [False, True])
def test_skewed_indices_cos_pair_is_exact(capi, hinted):
    """hot bins under the pair lock, with the sincos arithmetic between the claims (both the early adjoint and the stand-alone
    adjoint): A = C = 0, so u = 0, cos(u) = 1 exactly, and the sums are exact counts / exact sums of the integer-valued x --
    every lost or doubled update would show"""
    K, n = 1 << 17, (1 << 20) + 5
    rng = np.random.default_rng(4)
    Z = up(capi, np.zeros(K, np.float32))
    x = rng.integers(-2, 3, n).astype(np.float32)
    dx = up(capi, x)
    for idx in (np.full(n, 70001, np.uint32), (rng.integers(0, 100, n) + 5 * 8192).astype(np.uint32),
                np.where(rng.integers(0, 2, n) == 0, 3, rng.integers(0, K, n)).astype(np.uint32),
                (rng.integers(0, 2, n) * 8191 + 8192 * 3).astype(np.uint32)):
        di = up(capi, idx)
        b = capi.Bucketed("fmadd", Z, dx, Z, di, hints=capi.Bucketed.HINT_ADJOINT if hinted else 0)
        y = float(b.reduce("hsum", "sin", keep=True, keep_op="cos").numpy()[0])
        assert y == 0.0
        g1, gx = up(capi, np.zeros(K, np.float32)), up(capi, np.zeros(K, np.float32))
        b.scatter_add([gx, g1], [("cos", 0, True), ("cos", 0, False)])
        assert np.array_equal(g1.numpy().astype(np.float64), np.bincount(idx, minlength=K).astype(np.float64))
        assert np.array_equal(gx.numpy().astype(np.float64), np.bincount(idx, weights=x.astype(np.float64), minlength=K))
        b.destroy()


class _PartInfo(__import__("ctypes").Structure):
    import ctypes as _c
    _fields_ = [("shift", _c.c_int), ("n_buckets", _c.c_int), ("n", _c.c_size_t), ("range", _c.c_size_t),
                ("bucket_base", _c.c_void_p), ("local", _c.c_void_p),
                ("page_shift", _c.c_int), ("pages_full", _c.c_void_p), ("pages_part", _c.c_void_p), ("part_base", _c.c_void_p)]


@pytest.mark.parametrize("range_,shift", [(3000, 12), (1 << 20, 12), ((1 << 20) + 1, 14), (1 << 22, 14), ((1 << 22) + 5, 17),
                                           (1 << 25, 17), ((1 << 25) + 1, 19)])
@pytest.mark.parametrize("masked", [False, True])
def test_index_partition(capi, range_, shift, masked):
    """ek_hip_index_partition_*: the active entries of an index array grouped by bucket of the range (what
    enoki::vectorize_through runs on): every bucket holds exactly the bucket-local indices of its active entries (as a multiset:
    the order inside a bucket is unspecified), bucket_base is their exclusive prefix, the shift is the smallest of {12, 14, 17,
    19} with at most 256 buckets"""
    import ctypes
    n = 300007
    rng = np.random.default_rng(range_ % 1000 + masked)
    idx = rng.integers(0, range_, n).astype(np.uint32)
    idx[: n // 10] = idx[n // 10: 2 * (n // 10)]                   # duplicates
    mask = (rng.integers(0, 4, n) != 0).astype(np.uint8) if masked else None
    di = up(capi, idx)
    dm = up(capi, mask) if masked else None
    om = capi.operand(dm if masked else True, np.uint8)
    h = ctypes.c_void_p()
    capi.check(capi.lib.ek_hip_index_partition_create(capi.U32, ctypes.c_void_p(di.ptr), ctypes.byref(om), ctypes.c_size_t(n),
                                                      ctypes.c_size_t(range_), ctypes.byref(h)))
    info = _PartInfo()
    capi.check(capi.lib.ek_hip_index_partition_get(h, ctypes.byref(info)))
    assert info.shift == shift and info.n_buckets == -(-range_ // (1 << shift)) and info.n == n and info.range == range_
    assert info.page_shift == 0          # (below 2^20 entries: one contiguous run per bucket)
    base = capi.Buf(np.uint32, info.n_buckets + 1, own=False, ptr=info.bucket_base).numpy()
    active = idx[mask != 0] if masked else idx
    assert base[0] == 0 and base[-1] == active.size
    local = capi.Buf(np.uint32, max(int(base[-1]), 1), own=False, ptr=info.local).numpy()[: int(base[-1])]
    counts = np.bincount(active >> shift, minlength=info.n_buckets)
    assert np.array_equal(np.diff(base.astype(np.int64)), counts)
    # bucket by bucket: the same multiset of local indices
    order = np.argsort(active >> shift, kind="stable")
    want = (active[order] & ((1 << shift) - 1)).astype(np.uint32)
    for b in np.flatnonzero(counts)[:: max(1, info.n_buckets // 16)]:
        lo, hi = int(base[b]), int(base[b + 1])
        assert np.array_equal(np.sort(local[lo:hi]), np.sort(want[lo:hi])), b
    capi.check(capi.lib.ek_hip_index_partition_destroy(h))


@pytest.mark.parametrize("range_,shift", [((1 << 22) + 5, 17), (1 << 25, 17), ((1 << 25) + 1, 19), (1 << 19, 12)])
@pytest.mark.parametrize("masked", [False, True])
def test_index_partition_paged(capi, range_, shift, masked):
    """large inputs over at least 32 buckets go through the single-pass page partition (round 6): `local` holds pages of 2^page_shift
    bucket-local indices, a bucket owns a list of complete pages and a list of partially filled ones -- together exactly the
    multiset of its active entries' local indices"""
    import ctypes
    n = (1 << 21) + 4321
    rng = np.random.default_rng(range_ % 1000 + masked)
    idx = rng.integers(0, range_, n).astype(np.uint32)
    idx[: n // 10] = idx[n // 10: 2 * (n // 10)]                   # duplicates
    idx[5] = np.uint32(range_ + 3)                                  # out of range: dropped like a masked-out entry
    mask = (rng.integers(0, 4, n) != 0).astype(np.uint8) if masked else None
    di = up(capi, idx)
    dm = up(capi, mask) if masked else None
    om = capi.operand(dm if masked else True, np.uint8)
    h = ctypes.c_void_p()
    capi.check(capi.lib.ek_hip_index_partition_create(capi.U32, ctypes.c_void_p(di.ptr), ctypes.byref(om), ctypes.c_size_t(n),
                                                      ctypes.c_size_t(range_), ctypes.byref(h)))
    info = _PartInfo()
    capi.check(capi.lib.ek_hip_index_partition_get(h, ctypes.byref(info)))
    nb = info.n_buckets
    assert info.shift == shift and nb == -(-range_ // (1 << shift)) and info.page_shift in (5, 6), (info.shift, nb, info.page_shift)
    page = 1 << info.page_shift
    base = capi.Buf(np.uint32, nb + 1, own=False, ptr=info.bucket_base).numpy().astype(np.int64)
    pbase = capi.Buf(np.uint32, nb + 1, own=False, ptr=info.part_base).numpy().astype(np.int64)
    full = capi.Buf(np.uint32, max(int(base[-1]), 1), own=False, ptr=info.pages_full).numpy()
    part = capi.Buf(np.uint32, max(int(pbase[-1]), 1), own=False, ptr=info.pages_part).numpy()
    slots = int(max(full[: int(base[-1])].max(initial=0), (part[: int(pbase[-1])] >> 6).max(initial=0))) + 1
    local = capi.Buf(np.uint32, slots * page, own=False, ptr=info.local).numpy()
    keep = (idx < range_) & ((mask != 0) if masked else True)
    active = idx[keep]
    counts = np.bincount(active >> shift, minlength=nb)
    order = np.argsort(active >> shift, kind="stable")
    want = (active[order] & ((1 << shift) - 1)).astype(np.uint32)
    starts = np.concatenate([[0], np.cumsum(counts)])
    for b in range(nb):
        got = [local[int(pg) * page: int(pg) * page + page] for pg in full[base[b]: base[b + 1]]]
        got += [local[int(e >> 6) * page: int(e >> 6) * page + int(e & 63) + 1] for e in part[pbase[b]: pbase[b + 1]]]
        got = np.sort(np.concatenate(got)) if got else np.zeros(0, np.uint32)
        assert got.size == counts[b], (b, got.size, counts[b])
        if b % max(1, nb // 24) == 0:
            assert np.array_equal(got, np.sort(want[starts[b]: starts[b + 1]])), b
    capi.check(capi.lib.ek_hip_index_partition_destroy(h))


def test_index_partition_rejects_what_it_does_not_cover(capi):
    import ctypes
    di = up(capi, np.zeros(16, np.uint32))
    om = capi.operand(True, np.uint8)
    h = ctypes.c_void_p()
    for args in ((capi.U64, 16, 100), (capi.U32, 0, 100), (capi.U32, 16, 0), (capi.U32, 16, (256 << 19) + 1)):
        assert capi.lib.ek_hip_index_partition_create(args[0], ctypes.c_void_p(di.ptr), ctypes.byref(om), ctypes.c_size_t(args[1]),
                                                      ctypes.c_size_t(args[2]), ctypes.byref(h)) != 0
