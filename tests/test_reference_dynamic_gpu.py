"""Cases of the reference's tests/dynamic.cpp, tests/horiz.cpp and tests/memory.cpp that apply to a device array type,
re-expressed against enoki_amd.hip with the reference's own expected values (init, meshgrid layout, the haversine
GPS example with its 5918.18 km answer, even/odd compress, horizontal reductions on 1..N, masked gather/scatter)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ek():
    import enoki_amd.hip as m
    return m


def test04_init(ek):                                           # tests/dynamic.cpp:110-129
    v0, v1, v2 = ek.Float32.zero(11), ek.Float32.arange(11), ek.Float32.linspace(0, 1, 11)
    assert len(v0) == len(v1) == len(v2) == 11
    assert np.all(v0.numpy() == 0) and np.array_equal(v1.numpy(), np.arange(11, dtype=np.float32))
    assert np.abs(v2.numpy() - np.arange(11) / 10.0).max() < 1e-6


def test05_meshgrid(ek):                                       # tests/dynamic.cpp:131-153
    xy = ek.meshgrid(ek.Float32.linspace(0, 1, 2), ek.Float32.linspace(1, 4, 4))
    assert ek.slices(xy) == 8
    assert np.array_equal(xy.x.numpy(), [0, 1, 0, 1, 0, 1, 0, 1]) and np.array_equal(xy.y.numpy(), [1, 1, 2, 2, 3, 3, 4, 4])


def test06_haversine(ek):                                      # tests/dynamic.cpp:155-225
    n = 100
    lat1 = np.zeros(n, np.float32); lon1 = np.zeros(n, np.float32); lat2 = np.zeros(n, np.float32); lon2 = np.zeros(n, np.float32)
    lat1[0], lon1[0], lat2[0], lon2[0] = 51.5, 0.0, 38.8, -77.1
    reliable = np.zeros(n, np.uint8); reliable[0] = 1
    F = ek.Float32
    p1 = ek.Vector2f(F(lat1), F(lon1)); p2 = ek.Vector2f(F(lat2), F(lon2))
    deg_to_rad = F(np.pi / 180.0)
    d = (p2 - p1) * (deg_to_rad * F(0.5))
    s = ek.Vector2f(ek.sin(d.x), ek.sin(d.y))
    s = s * s
    a = s.x + s.y * ek.cos(p1.x * deg_to_rad) * ek.cos(p2.x * deg_to_rad)
    dist = ek.select(ek.Mask(reliable), F(6371.0 * 2.0) * ek.atan2(ek.sqrt(a), ek.sqrt(F(1.0) - a)), F(float("nan")))
    out = dist.numpy()
    assert abs(out[0] - 5918.18) < 1e-2 and np.all(np.isnan(out[1:]))


def test07_compress(ek):                                       # tests/dynamic.cpp:226-272 (even entries, then odd entries)
    n = 64
    i = ek.UInt32.arange(n)
    x = ek.Float32(i); y = ek.Float32(i * ek.UInt32(100))
    even = ((i >> ek.UInt32(1)) << ek.UInt32(1)) == i
    odd = ((i >> ek.UInt32(1)) << ek.UInt32(1)) != i
    ex, ox = ek.compress(x, even).numpy(), ek.compress(x, odd).numpy()
    ey, oy = ek.compress(y, even).numpy(), ek.compress(y, odd).numpy()
    assert np.array_equal(ex, np.arange(0, n, 2)) and np.array_equal(ox, np.arange(1, n, 2))
    assert np.array_equal(ey, 100.0 * np.arange(0, n, 2)) and np.array_equal(oy, 100.0 * np.arange(1, n, 2))


@pytest.mark.parametrize("n", [1, 2, 3, 4, 8, 16, 31, 32, 1000])
def test_horiz(ek, n):                                         # tests/horiz.cpp:16-170 on the sample 1..n
    for cls, dt in ((ek.Float32, np.float32), (ek.Int32, np.int32), (ek.UInt32, np.uint32), (ek.Float64, np.float64),
                    (ek.Int64, np.int64)):
        v = np.arange(1, n + 1).astype(dt)
        a = cls(v)
        assert ek.hsum(a).numpy()[0] == v.sum(dtype=dt)
        if n <= 12:
            assert ek.hprod(a).numpy()[0] == v.prod(dtype=dt)
        assert ek.hmin(a).numpy()[0] == 1 and ek.hmax(a).numpy()[0] == n
        m = a > cls(dt(n // 2))
        assert ek.count(m) == n - n // 2 and ek.any(m) == (n - n // 2 > 0) and ek.all(m) == (n // 2 == 0)
        assert ek.none(m) == (n - n // 2 == 0) if hasattr(ek, "none") else True
    a = ek.Vector3f(ek.Float32(np.arange(n, dtype=np.float32)), ek.Float32(2.0), ek.Float32(-1.0))
    b = ek.Vector3f(ek.Float32(1.0), ek.Float32(np.arange(n, dtype=np.float32)), ek.Float32(3.0))
    assert np.array_equal(ek.dot(a, b).numpy(), np.arange(n, dtype=np.float32) * 3 - 3)      # test09_dot


def test_memory_gather_scatter_masked(ek):                     # tests/memory.cpp:47-200, tests/memory2.cpp:302-320
    n = 1024
    mem = np.arange(n, dtype=np.float32) * 2
    src = ek.Float32(mem)
    idx_np = ((np.arange(n, dtype=np.uint32) * 7919) % n).astype(np.uint32)
    idx = ek.UInt32(idx_np)
    assert np.array_equal(ek.gather(src, idx).numpy(), mem[idx_np])
    mask_np = (np.arange(n) % 2 == 0)
    got = ek.gather(src, idx, ek.Mask(mask_np.astype(np.uint8))).numpy()
    assert np.array_equal(got, np.where(mask_np, mem[idx_np], 0))                            # masked lanes read as zero
    dst = ek.Float32.zero(n)
    ek.scatter(dst, src, idx)                                                                # idx is a permutation
    want = np.zeros(n, np.float32); want[idx_np] = mem
    assert np.array_equal(dst.numpy(), want)
    dst = ek.Float32.full(-1.0, n)
    ek.scatter(dst, src, idx, ek.Mask(mask_np.astype(np.uint8)))
    want = np.full(n, -1.0, np.float32); want[idx_np[mask_np]] = mem[mask_np]
    assert np.array_equal(dst.numpy(), want)
    # integer scatter_add with duplicates is exact
    acc = ek.UInt32.zero(16)
    ek.scatter_add(acc, ek.UInt32(np.ones(n, np.uint32)), ek.UInt32((np.arange(n) % 16).astype(np.uint32)))
    assert np.all(acc.numpy() == n // 16)
