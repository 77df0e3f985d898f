"""Step graphs (ek_hip_graph_*): the launches of one forward + backward() are captured once and replayed without host work;
the replay must produce what the eager step produces -- also after the INPUT BUFFERS were refilled in place."""
import ctypes

import numpy as np
import pytest

from conftest import bits_equal, cfg3b_truth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ek():
    import enoki_amd.hip_autodiff as m
    m.hip_init(0)
    return m


def _refill(capi, arr, host):
    host = np.ascontiguousarray(host)
    capi.check(capi.lib.ek_hip_memcpy_to_device(ctypes.c_void_p(arr.data_ptr()), host.ctypes.data_as(ctypes.c_void_p),
                                                ctypes.c_size_t(host.nbytes)))


@pytest.mark.parametrize("n,K", [(300007, 65536), (1 << 22, 1 << 20)])
def test_cfg3b_step_graph_replay(ek, capi, n, K):
    rng = np.random.default_rng(n)
    hA, hB = rng.uniform(-1, 1, K).astype(np.float32), rng.uniform(-1, 1, K).astype(np.float32)
    hx = rng.uniform(-1, 1, n).astype(np.float32); hidx = rng.integers(0, K, n).astype(np.uint32)
    A0, B0, x, idx = ek.Float32(hA), ek.Float32(hB), ek.Float32(hx), ek.UInt32(hidx)
    out = {}

    def step():
        A, B = ek.Float32(A0), ek.Float32(B0)
        ek.set_requires_gradient(A); ek.set_requires_gradient(B)
        y = ek.hsum(ek.sin(ek.fmadd(ek.gather(A, idx), x, ek.gather(B, idx))))
        ek.backward(y)
        out["y"], out["gA"], out["gB"] = ek.detach(y), ek.gradient(A), ek.gradient(B)

    step()                                              # eager: warms the allocator, gives the reference values
    eager = (out["y"].numpy().copy(), out["gA"].numpy().copy(), out["gB"].numpy().copy())
    launches0 = ek.hip_launch_count()
    ek.hip_graph_begin()
    step()
    g = ek.hip_graph_end()
    try:
        per_step = ek.hip_graph_launch_count(g)
        assert per_step >= 4 and ek.hip_launch_count() - launches0 == per_step       # (round 4: single-pass partition, fewer launches)
        t = cfg3b_truth(hA, hB, hx, hidx)
        for _ in range(3):
            ek.hip_graph_launch(g)
            y, gA, gB = out["y"].numpy(), out["gA"].numpy(), out["gB"].numpy()
            # (the bucket-ordered forward sums in the order its partition happened to produce: class D, not run-to-run
            # reproducible -- ENOKI_HIP_DETERMINISTIC=1 is the reproducible mode)
            assert abs(float(y[0]) - t["y"]) <= t["y_bound"] and abs(float(y[0]) - float(eager[0][0])) <= 2 * t["y_bound"]
            assert np.all(np.abs(gA - t["gA"]) <= t["gA_bound"]) and np.all(np.abs(gB - t["gB"]) <= t["gB_bound"])
        # new contents in the SAME buffers: the graph reads the new inputs
        hA2, hx2 = rng.uniform(-1, 1, K).astype(np.float32), rng.uniform(-1, 1, n).astype(np.float32)
        _refill(capi, A0, hA2); _refill(capi, x, hx2)
        ek.hip_graph_launch(g)
        t2 = cfg3b_truth(hA2, hB, hx2, hidx)
        assert abs(float(out["y"].numpy()[0]) - t2["y"]) <= t2["y_bound"]
        assert np.all(np.abs(out["gA"].numpy() - t2["gA"]) <= t2["gA_bound"])
        # eager work after the capture must not disturb the graph's buffers
        junk = [ek.Float32(np.full(n, 3.0, np.float32)) * ek.Float32(2.0) for _ in range(4)]
        ek.hip_graph_launch(g)
        assert np.all(np.abs(out["gB"].numpy() - t2["gB"]) <= t2["gB_bound"])
        del junk
    finally:
        ek.hip_graph_destroy(g)
    # reads to the host are refused inside a capture, and the library recovers
    ek.hip_graph_begin()
    with pytest.raises(RuntimeError):
        ek.count(x > ek.Float32(0.0))
    g2 = ek.hip_graph_end()
    ek.hip_graph_destroy(g2)
    step()
    # two bucket-ordered evaluations of the same sum: same terms, an order that depends on the partition (class D)
    again = ek.hsum(ek.sin(ek.fmadd(ek.gather(A0, idx), x, ek.gather(B0, idx)))).numpy()
    assert abs(float(out["y"].numpy()[0]) - t2["y"]) <= t2["y_bound"] and abs(float(again[0]) - t2["y"]) <= t2["y_bound"]


def test_host_waits_are_refused_inside_a_capture(ek):
    """everything that makes the host wait for the device -- hip_sync, the read-back of the deterministic scatter_add,
    host -> device uploads -- fails with a clear message while a step graph is being captured and leaves BOTH the capture
    and the library usable (a hipStreamSynchronize on a capturing stream would invalidate the capture and poison every
    later launch: the state `bench.py --deterministic` ran into)"""
    rng = np.random.default_rng(2)
    n, K = 1 << 18, 1 << 16
    v = ek.Float32(rng.standard_normal(n).astype(np.float32))
    idx = ek.UInt32(rng.integers(0, K, n).astype(np.uint32))
    ek.hip_sync()
    ek.hip_graph_begin()
    try:
        with pytest.raises(RuntimeError, match="captured step graph"):
            ek.hip_sync()
        ek.hip_set_tuning("deterministic", 1)
        try:
            t = ek.Float32.zero(K)
            with pytest.raises(RuntimeError, match="captured step graph"):
                ek.scatter_add(t, v, idx, idx < ek.UInt32(K // 2))   # a mask ARRAY: the number of active pairs is read back
            td = ek.Float32.zero(K)
            ek.scatter_add(td, v, idx)                               # no mask array: nothing to read back, recorded
        finally:
            ek.hip_set_tuning("deterministic", 0)
        with pytest.raises(RuntimeError, match="captured step graph"):
            ek.Float32(np.ones(16, np.float32))                  # host -> device upload
        w = v * ek.Float32(2.0)                                     # ordinary work is still recorded
    finally:
        g = ek.hip_graph_end()
    ek.hip_graph_launch(g)
    assert bits_equal(w.numpy(), v.numpy() * np.float32(2))
    # the deterministic scatter_add inside the graph: element-order sums, bit for bit
    want = np.zeros(K, np.float32)
    np.add.at(want, idx.numpy(), v.numpy())
    assert bits_equal(td.numpy(), want)
    ek.hip_graph_destroy(g)
    # and the eager library is alive
    t = ek.Float32.zero(K)
    ek.scatter_add(t, v, idx)
    assert np.allclose(t.numpy(), np.bincount(idx.numpy(), weights=v.numpy().astype(np.float64), minlength=K), atol=1e-3)


def test_arrays_deferred_before_a_capture_do_not_live_in_the_graph_pool(ek):
    """an unevaluated result that predates a capture is evaluated by hip_graph_begin(): if it were evaluated INSIDE the capture
    its storage would come from the graph's private pool and dangle once the graph is destroyed"""
    n = 1 << 18
    x = ek.Float32.linspace(0.0, 3.0, n)
    s = ek.sin(x)                                   # deferred (n >= 64 Ki): no storage yet
    l0 = ek.hip_launch_count()
    ek.hip_graph_begin()
    assert ek.hip_launch_count() - l0 == 1          # evaluated by hip_graph_begin, outside the capture
    y = s * ek.Float32(2.0)
    g = ek.hip_graph_end()
    ek.hip_graph_launch(g)
    want = np.sin(np.linspace(0.0, 3.0, n, dtype=np.float32).astype(np.float64))
    assert np.allclose(y.numpy(), 2 * want, atol=1e-6)
    ek.hip_graph_destroy(g)
    junk = [ek.Float32.zero(n) for _ in range(8)]   # would reuse the pool's blocks
    assert np.allclose(s.numpy(), want, atol=1e-6)
    del junk

