"""Struct gathers through staged records (ek_hip_gather_multi_sized, memory.hip k_stage_records / k_gather_records): the
components of a structure of arrays are interleaved into 8- or 16-byte records and looked up with ONE request per
element.  Results must equal the per-component gather (cuda.h:845-864 via array_struct.h:9-40) bit for bit -- these are
moves of bit patterns -- for every component count, element width, index type, mask kind and alignment."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ek():
    from enoki_amd import capi
    capi.init()
    yield capi
    capi.set_tuning("gather_records", 1)


def tables_of(rng, dtype, count, k):
    if np.dtype(dtype).kind == "f":
        t = [rng.standard_normal(k).astype(dtype) for _ in range(count)]
        for a in t:                                   # bit patterns must survive: NaN payloads, -0, denormals
            a[:4] = np.array([np.nan, -0.0, np.finfo(dtype).tiny / 4, np.inf], dtype=dtype)
        return t
    return [rng.integers(0, np.iinfo(dtype).max, k, dtype=dtype) for _ in range(count)]


@pytest.mark.parametrize("mode", [2, 0])
@pytest.mark.parametrize("dtype,count", [(np.float32, 2), (np.float32, 3), (np.float32, 4), (np.uint32, 3), (np.float64, 2),
                                         (np.int64, 2), (np.float64, 3)])
@pytest.mark.parametrize("index_dtype", [np.uint32, np.int32, np.int64])
def test_struct_gather_bit_exact(ek, mode, dtype, count, index_dtype):
    ek.set_tuning("gather_records", mode)
    rng = np.random.default_rng(hash((mode, np.dtype(dtype).name, count, np.dtype(index_dtype).name)) & 0xffff)
    for k, n in [(1000, 5), (1000, 4099), (70001, 100003), (5, 1 << 16)]:
        host = tables_of(rng, dtype, count, k)
        idx = rng.integers(0, k, n).astype(index_dtype)
        msk = (rng.random(n) < 0.7).astype(np.uint8)
        dev = [ek.Buf.from_numpy(t) for t in host]
        di, dm = ek.Buf.from_numpy(idx), ek.Buf.from_numpy(msk)
        view = {4: np.uint32, 8: np.uint64}[np.dtype(dtype).itemsize]
        for mask, hm in ((True, np.ones(n, bool)), (dm, msk.astype(bool)), (False, np.zeros(n, bool))):
            outs = ek.gather_multi(dev, di, mask)
            for c in range(count):
                expect = np.where(hm, host[c][idx], np.zeros(1, dtype)).astype(dtype)
                assert np.array_equal(outs[c].numpy().view(view), expect.view(view)), (k, n, c)


def test_unaligned_tables_and_outputs(ek):
    """tables that start 4 bytes into an allocation take the scalar staging body; same results"""
    ek.set_tuning("gather_records", 2)
    rng = np.random.default_rng(3)
    k, n = 4097, 9001
    host = [rng.standard_normal(k + 1).astype(np.float32) for _ in range(3)]
    keep = [ek.Buf.from_numpy(t) for t in host]
    dev = [b.view(1, k) for b in keep]
    idx = rng.integers(0, k, n).astype(np.uint32)
    outs = ek.gather_multi(dev, ek.Buf.from_numpy(idx))
    for c in range(3):
        assert np.array_equal(outs[c].numpy(), host[c][1:][idx])
    del keep


def test_size_policy_picks_records_for_large_tables(ek):
    """default policy: tables beyond the L2 with enough lookups -> staged records; small tables -> plain kernels"""
    ek.set_tuning("gather_records", 1)
    rng = np.random.default_rng(5)
    for k, n, expect_records in [(1 << 22, 1 << 23, True), (1 << 20, 1 << 23, False), (1 << 12, 1 << 22, False), (1 << 22, 1 << 10, False)]:
        host = [rng.standard_normal(k).astype(np.float32) for _ in range(3)]
        dev = [ek.Buf.from_numpy(t) for t in host]
        idx = rng.integers(0, k, n).astype(np.uint32)
        di = ek.Buf.from_numpy(idx)
        ek.profile_begin()
        outs = ek.gather_multi(dev, di)
        names = {p["kernel"] for p in ek.profile_end()}
        assert ("gather_records" in names) == expect_records, (k, n, names)
        for c in range(3):
            assert np.array_equal(outs[c].numpy(), host[c][idx])


def test_vector3f_gather_through_the_binding(ek):
    """gather<Vector3f>(soa, index) of the python surface goes through the same entry point"""
    import enoki_amd.hip as eh
    rng = np.random.default_rng(9)
    k, n = 1 << 22, 1 << 23
    comps = [rng.standard_normal(k).astype(np.float32) for _ in range(3)]
    v = eh.Vector3f(*[eh.Float32(c) for c in comps])
    idx = rng.integers(0, k, n).astype(np.uint32)
    ek.profile_begin()
    r = eh.gather(v, eh.UInt32(idx))
    names = {p["kernel"] for p in ek.profile_end()}
    assert "gather_records" in names, names
    for c, got in enumerate((r.x, r.y, r.z)):
        assert np.array_equal(got.numpy(), comps[c][idx])


def test_differentiable_struct_gather(ek):
    """gather<Vector3fD>(texture, index) with large component tables: ONE record lookup per element for the values, the tape
    still gets a gather node per component (adjoint = one multi-table scatter_add); values bit-identical to the
    per-component path, gradients exact (integer-valued data)"""
    import enoki_amd.hip_autodiff as ed
    rng = np.random.default_rng(11)
    k, n = 1 << 22, 1 << 23
    comps = [rng.integers(-8, 9, k).astype(np.float32) for _ in range(3)]
    wts = [rng.integers(-3, 4, n).astype(np.float32) for _ in range(3)]
    idx = rng.integers(0, k, n).astype(np.uint32)
    msk = rng.random(n) < 0.8
    results = {}
    for mode in (1, 0):
        ek.set_tuning("gather_records", mode)
        leaves = [ed.Float32(c) for c in comps]
        for c in leaves:
            ed.set_requires_gradient(c)
        tex = ed.Vector3f(*leaves)
        ek.profile_begin()
        got = ed.gather(tex, ed.UInt32(idx), ed.Mask(msk))
        names = {p["kernel"] for p in ek.profile_end()}
        assert ("gather_records" in names) == (mode == 1), names
        loss = ed.hsum(got.x * ed.Float32(wts[0]) + got.y * ed.Float32(wts[1]) + got.z * ed.Float32(wts[2]))
        ed.backward(loss)
        results[mode] = ([ed.detach(v).numpy() for v in (got.x, got.y, got.z)],
                         [ed.gradient(c).numpy() for c in leaves])
    ek.set_tuning("gather_records", 1)
    for c in range(3):
        expect = np.where(msk, comps[c][idx], np.float32(0))
        assert np.array_equal(results[1][0][c], expect) and np.array_equal(results[0][0][c], expect)
        grad = np.bincount(idx[msk], weights=wts[c][msk].astype(np.float64), minlength=k).astype(np.float32)
        assert np.array_equal(results[1][1][c], grad), c
        assert np.array_equal(results[0][1][c], grad), c
