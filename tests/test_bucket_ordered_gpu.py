"""HIPArray / DiffArray level: `fmadd(gather(A, idx), x, gather(B, idx))` stays unevaluated one step longer than its gathers
(a kind-2 deferred node, include/enoki/hip.h) and is consumed BUCKET BY BUCKET when the consumer does not care about the
element order -- the forward `hsum(sin(u))` and the adjoint scatter_add of BASELINE config 3b.  Arrays stay immutable
values: whatever else touches u, its tables, x or the index array sees exactly what eager evaluation gives.

  * element values (u, cos(u), ...) are BIT-IDENTICAL to eager evaluation however they are forced;
  * reductions and gradients are the same multisets summed in another order: class D, checked against float64 with the
    bound of our summation depth and EXACTLY on integer-valued data;
  * deterministic mode never takes the bucket-ordered path.
"""
import ctypes
import json

import numpy as np
import pytest

from conftest import bits_equal, cfg3b_truth

pytestmark = pytest.mark.gpu

N, K = (1 << 20) + 4099, (1 << 18) + 17         # 17 buckets, ragged ends


@pytest.fixture(scope="module")
def ek():
    import enoki_amd.hip as m
    m.hip_init(0)
    return m


@pytest.fixture(scope="module")
def ad():
    import enoki_amd.hip_autodiff as m
    m.hip_init(0)
    return m


def data(seed=5, integer=False):
    rng = np.random.default_rng(seed)
    if integer:
        A = rng.integers(-3, 4, K).astype(np.float32); B = rng.integers(-3, 4, K).astype(np.float32)
        x = rng.integers(-2, 3, N).astype(np.float32)
    else:
        A = rng.uniform(-1, 1, K).astype(np.float32); B = rng.uniform(-1, 1, K).astype(np.float32)
        x = rng.uniform(-1, 1, N).astype(np.float32)
    return A, B, x, rng.integers(0, K, N).astype(np.uint32)


def kernels(m, fn):
    m.hip_profile_begin()
    r = fn()
    prof = json.loads(m.hip_profile_end())
    return r, {k["kernel"]: k["launches"] for k in prof if k["launches"]}


def eager_u(ek, A, B, x, idx):
    ek.hip_set_defer(False)
    try:
        return ek.fmadd(ek.gather(ek.Float32(A), ek.UInt32(idx)), ek.Float32(x), ek.gather(ek.Float32(B), ek.UInt32(idx))).numpy()
    finally:
        ek.hip_set_defer(True)


def test_bucket_ordered_semantics(ek, capi):
    A, B, x, idx = data()
    u_ref = eager_u(ek, A, B, x, idx)
    dA, dB, dx, di = ek.Float32(A), ek.Float32(B), ek.Float32(x), ek.UInt32(idx)
    make = lambda: ek.fmadd(ek.gather(dA, di), dx, ek.gather(dB, di))
    t = cfg3b_truth(A, B, x, idx)

    # consumed by a reduction through a unary map: bucket order, no element-order kernel
    y, ks = kernels(ek, lambda: ek.hsum(ek.sin(make())))
    assert "bucket_pair_fma_reduce" in ks and "bucket_partition" in ks and not any(k.startswith("gather") for k in ks), ks
    assert abs(float(y.numpy()[0]) - t["y"]) <= t["y_bound"]
    # statistically far inside the worst case: chains of <= 64 additions, then trees
    assert abs(float(y.numpy()[0]) - t["y"]) <= t["y_stat_bound"]
    # order-independent reductions are bit-exact
    assert bits_equal(ek.hmax(make()).numpy(), np.array([u_ref.max()]))
    assert bits_equal(ek.hmin(ek.abs(make())).numpy(), np.array([np.abs(u_ref).min()]))

    # forced via numpy() / data_ptr() after a bucket-ordered reduction: bit-identical to eager
    u = make()
    ek.hsum(ek.cos(u))
    assert bits_equal(u.numpy(), u_ref)
    u = make()
    ek.hsum(u)
    assert u.data_ptr() != 0 and bits_equal(u.numpy(), u_ref)
    # second, element-order consumer
    u = make()
    s = ek.sin(u)
    y1 = ek.hsum(s)
    w, ks = kernels(ek, lambda: u * dx)
    assert bits_equal(w.numpy(), u_ref * x) and "gather_pair_fmadd" in ks
    assert bits_equal(s.numpy(), ek.sin(ek.Float32(u_ref)).numpy())          # the map built on the unevaluated u
    assert abs(float(y1.numpy()[0]) - t["y"]) <= t["y_bound"]
    # mutated table / x / index while u is pending: u holds the OLD contents
    T = ek.Float32(A)
    u = ek.fmadd(ek.gather(T, di), dx, ek.gather(dB, di))
    ek.hsum(ek.sin(u))
    ek.scatter(T, ek.Float32(np.full(K, 9.0, np.float32)), ek.UInt32.arange(K))
    assert bits_equal(u.numpy(), u_ref) and np.all(T.numpy() == 9.0)
    X = ek.Float32(x)
    u = ek.fmadd(ek.gather(dA, di), X, ek.gather(dB, di))
    ek.scatter(X, ek.Float32(np.zeros(N, np.float32)), ek.UInt32.arange(N))
    assert bits_equal(u.numpy(), u_ref)
    I = ek.UInt32(idx)
    u = ek.fmadd(ek.gather(dA, I), dx, ek.gather(dB, I))
    ek.scatter(I, ek.UInt32(np.zeros(N, np.uint32)), ek.UInt32.arange(N))
    assert bits_equal(u.numpy(), u_ref)
    # the source handles go away: the node keeps the buffers
    u = ek.fmadd(ek.gather(ek.Float32(A), ek.UInt32(idx)), ek.Float32(x), ek.gather(ek.Float32(B), ek.UInt32(idx)))
    assert abs(float(ek.hsum(ek.sin(u)).numpy()[0]) - t["y"]) <= t["y_bound"] and bits_equal(u.numpy(), u_ref)
    # the other members of the family
    for name in ("fmsub", "fnmadd", "fnmsub"):
        f = getattr(ek, name)
        ek.hip_set_defer(False)
        try:
            want = f(ek.gather(dA, di), dx, ek.gather(dB, di)).numpy()
        finally:
            ek.hip_set_defer(True)
        got = f(ek.gather(dA, di), dx, ek.gather(dB, di))
        assert bits_equal(ek.hmax(got).numpy(), np.array([want.max()])), name       # bucket order
        assert bits_equal(got.numpy(), want), name                                  # then forced in element order
    # switched off: element order
    ek.hip_set_tuning("bucket_ordered", 0)
    try:
        _, ks = kernels(ek, lambda: ek.hsum(ek.sin(make())))
        assert any(k.startswith("gather") for k in ks) and not any(k.startswith("bucket") for k in ks), ks
    finally:
        ek.hip_set_tuning("bucket_ordered", 1)


def test_backward_reuses_the_partition(ad):
    A, B, x, idx = data(seed=7)
    t = cfg3b_truth(A, B, x, idx)

    def step(A_, B_, x_, idx_):
        dA, dB = ad.Float32(A_), ad.Float32(B_)
        ad.set_requires_gradient(dA); ad.set_requires_gradient(dB)
        di = ad.UInt32(idx_)
        y = ad.hsum(ad.sin(ad.fmadd(ad.gather(dA, di), ad.Float32(x_), ad.gather(dB, di))))
        ad.backward(y)
        return ad.detach(y).numpy(), ad.gradient(dA).numpy(), ad.gradient(dB).numpy()

    (y, gA, gB), ks = kernels(ad, lambda: step(A, B, x, idx))
    # ONE count / scan / partition per step, in the forward; hsum(sin(u)) with cos(u) still held by the tape is the shape of a
    # derivative: the sums of cos(u) and x cos(u) per table entry are formed in the SAME pass (EK_BUCKETED_HINT_ADJOINT) and
    # the backward sweep only folds them
    assert ks.get("bucket_partition") == 1 and ks.get("bucket_directory") == 1 and ks.get("bucket_pair_fma_reduce_adjoint") == 1, ks
    assert "bucket_count" not in ks and "bucket_scan" not in ks, ks          # single-pass partition (round 4): no count pass, no scans
    assert ks.get("scatter_add_fold") == 1, ks
    assert not any(k in ks for k in ("scatter_add_partition", "scatter_add_count", "gather_pair_fmadd", "sincos", "hsum_map",
                                     "bucket_accumulate", "bucket_pair_fma_reduce")), ks
    # a seed other than 1 is not what was summed (0.5 * cos(u) is an evaluated array): the ordinary scatter_add, same values
    def step_scaled():
        dA, dB = ad.Float32(A), ad.Float32(B)
        ad.set_requires_gradient(dA); ad.set_requires_gradient(dB)
        di = ad.UInt32(idx)
        y = ad.hsum(ad.sin(ad.fmadd(ad.gather(dA, di), ad.Float32(x), ad.gather(dB, di)))) * ad.Float32(0.5)
        ad.backward(y)
        return ad.gradient(dA).numpy(), ad.gradient(dB).numpy()
    (hA, hB), ks2 = kernels(ad, step_scaled)
    assert ks2.get("bucket_partition") == 1, ks2
    assert np.all(np.abs(hA - 0.5 * t["gA"]) <= t["gA_bound"]) and np.all(np.abs(hB - 0.5 * t["gB"]) <= t["gB_bound"])
    # switched off: forward and adjoint are two passes again
    ad.hip_set_tuning("early_adjoint", 0)
    try:
        (y0, gA0, gB0), ks0 = kernels(ad, lambda: step(A, B, x, idx))
        assert ks0.get("bucket_accumulate") == 1 and ks0.get("bucket_pair_fma_reduce") == 1, ks0
        assert np.all(np.abs(gA0 - t["gA"]) <= t["gA_bound"]) and np.all(np.abs(gB0 - t["gB"]) <= t["gB_bound"])
    finally:
        ad.hip_set_tuning("early_adjoint", 1)
    assert abs(float(y[0]) - t["y"]) <= t["y_bound"]
    assert np.all(np.abs(gA - t["gA"]) <= t["gA_bound"]) and np.all(np.abs(gB - t["gB"]) <= t["gB_bound"])
    # integer-valued data: sin / cos are not exact, so take the exact part -- y = hsum(u), grads x and 1 -- via a linear loss
    Ai, Bi, xi, ii = data(seed=8, integer=True)
    dA, dB = ad.Float32(Ai), ad.Float32(Bi)
    ad.set_requires_gradient(dA); ad.set_requires_gradient(dB)
    di = ad.UInt32(ii)
    y = ad.hsum(ad.fmadd(ad.gather(dA, di), ad.Float32(xi), ad.gather(dB, di)))
    ad.backward(y)
    u = Ai[ii].astype(np.float64) * xi + Bi[ii]
    assert float(ad.detach(y).numpy()[0]) == u.sum()
    assert np.array_equal(ad.gradient(dA).numpy(), np.bincount(ii, weights=xi.astype(np.float64), minlength=K).astype(np.float32))
    assert np.array_equal(ad.gradient(dB).numpy(), np.bincount(ii, minlength=K).astype(np.float32))
    # deterministic mode: element-order kernels, gradients bit-identical to the CPU order
    ad.hip_set_tuning("deterministic", 1)
    try:
        (yd, gAd, gBd), ks = kernels(ad, lambda: step(A, B, x, idx))
        assert not any(k.startswith("bucket") for k in ks), ks
        import enoki_amd.hip as ekc
        cu = ad.cos(ad.Float32(eager_u(ekc, A, B, x, idx))).numpy()
        want = np.zeros(K, np.float32)
        np.add.at(want, idx, cu)
        assert bits_equal(gBd, want)
    finally:
        ad.hip_set_tuning("deterministic", 0)


def test_bucket_ordered_step_graph(ad, capi):
    """the captured step contains the bucket-ordered kernels; replays follow refilled inputs; arrays that are still
    unevaluated when the capture ends are evaluated inside the graph (and refreshed by every replay)"""
    A, B, x, idx = data(seed=9)
    import enoki_amd.hip as ekc
    A0, B0, X, I = ad.Float32(A), ad.Float32(B), ad.Float32(x), ad.UInt32(idx)
    out = {}

    def step():
        dA, dB = ad.Float32(A0), ad.Float32(B0)
        ad.set_requires_gradient(dA); ad.set_requires_gradient(dB)
        u = ad.fmadd(ad.gather(dA, I), X, ad.gather(dB, I))
        y = ad.hsum(ad.sin(u))
        ad.backward(y)
        out["y"], out["gA"], out["gB"] = ad.detach(y), ad.gradient(dA), ad.gradient(dB)
        # pending at the end of the capture (a fusable unary result of N elements)
        out["e"] = ekc.exp(ad.detach(y) * ad.detach(X))

    step()
    ad.hip_sync()
    ad.hip_graph_begin()
    try:
        step()
    finally:
        g = ad.hip_graph_end()
    try:
        for hA, hx in ((A, x), (np.roll(A, 3), -x)):
            for arr, host in ((A0, hA), (X, hx)):
                host = np.ascontiguousarray(host)
                capi.check(capi.lib.ek_hip_memcpy_to_device(ctypes.c_void_p(arr.data_ptr()), host.ctypes.data_as(ctypes.c_void_p),
                                                            ctypes.c_size_t(host.nbytes)))
            ad.hip_graph_launch(g)
            t = cfg3b_truth(hA, B, hx, idx)
            y = out["y"].numpy()
            assert abs(float(y[0]) - t["y"]) <= t["y_bound"]
            assert np.all(np.abs(out["gA"].numpy() - t["gA"]) <= t["gA_bound"]) and np.all(np.abs(out["gB"].numpy() - t["gB"]) <= t["gB_bound"])
            assert bits_equal(out["e"].numpy(), ad.exp(ad.Float32(y) * ad.Float32(hx)).numpy())
    finally:
        ad.hip_graph_destroy(g)


def test_x_needs_a_gradient_too(ad):
    """x on the tape as well: its adjoint cos(u) * A[idx] is an ELEMENT-order product, so the held cos(u) and u are evaluated in
    element order after the bucket-ordered forward pass (whose gradient sums are dropped with the partition): three gradients,
    all within their class-D bounds; and once more with x the only differentiated input"""
    A, B, x, idx = data(seed=11)
    t = cfg3b_truth(A, B, x, idx)
    ii = idx.astype(np.int64)
    u = A.astype(np.float64)[ii] * x + B.astype(np.float64)[ii]
    gx_ref = np.cos(u) * A.astype(np.float64)[ii]
    for tables in (True, False):
        dA, dB, dx = ad.Float32(A), ad.Float32(B), ad.Float32(x)
        if tables:
            ad.set_requires_gradient(dA); ad.set_requires_gradient(dB)
        ad.set_requires_gradient(dx)
        di = ad.UInt32(idx)
        y = ad.hsum(ad.sin(ad.fmadd(ad.gather(dA, di), dx, ad.gather(dB, di))))
        ad.backward(y)
        assert abs(float(ad.detach(y).numpy()[0]) - t["y"]) <= t["y_bound"]
        assert np.all(np.abs(ad.gradient(dx).numpy() - gx_ref) <= 8 * 2.0 ** -24 * (1.0 + np.abs(gx_ref)))
        if tables:
            assert np.all(np.abs(ad.gradient(dA).numpy() - t["gA"]) <= t["gA_bound"])
            assert np.all(np.abs(ad.gradient(dB).numpy() - t["gB"]) <= t["gB_bound"])
