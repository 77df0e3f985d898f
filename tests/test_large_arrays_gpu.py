"""Maximum-size edge case: arrays with more than 2^31 elements (8 GiB of 4-byte elements, a small fraction of the
288 GB of an MI355X).  Checks 64-bit indexing in the streaming kernels, reductions, casts, gather and scatter through
size-independent properties (closed-form sums, counts, spot values)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
N = (1 << 31) + 7


@pytest.fixture(scope="module")
def ek():
    import enoki_amd.hip as m
    return m


def test_beyond_2_31_elements(ek):
    x = ek.UInt32.arange(N)                                     # 0 .. 2^31 + 6
    assert len(x) == N and x[N - 1] == N - 1 and x[(1 << 31)] == (1 << 31)
    y = x + x                                                   # wraps modulo 2^32 in the upper half
    assert y[N - 1] == (2 * (N - 1)) % (1 << 32) and y[12345] == 24690
    hi = x >= ek.UInt32(1 << 31)
    assert ek.count(hi) == 7 and ek.any(hi) and not ek.all(hi)
    assert ek.count(x >= ek.UInt32(0)) == N                     # count is carried in 64 bits
    del y, hi
    s = ek.hsum(ek.UInt64(x)).numpy()[0]                        # cast to 16 GiB of u64, exact closed form
    assert int(s) == N * (N - 1) // 2
    assert ek.hmax(x).numpy()[0] == N - 1 and ek.hmin(x).numpy()[0] == 0
    # gather / scatter at element offsets whose BYTE offset does not fit 32 bits
    idx_np = np.array([N - 1, N - 5, 1 << 31, (1 << 30) + 3, 5], np.uint32)
    idx = ek.UInt32(idx_np)
    assert np.array_equal(ek.gather(x, idx).numpy(), idx_np)
    ek.scatter(x, ek.UInt32(np.arange(5, dtype=np.uint32) + 100), idx)
    assert [x[int(i)] for i in idx_np] == [100, 101, 102, 103, 104]
    ek.scatter_add(x, ek.UInt32(np.full(5, 7, np.uint32)), idx)
    assert [x[int(i)] for i in idx_np] == [107, 108, 109, 110, 111]
    del x
    # float path: fmadd over > 2^31 elements, checked at the ends and through an exact reduction of small integers
    f = ek.Float32.full(0.5, N)
    g = ek.fmadd(f, ek.Float32(2.0), ek.Float32(1.0))           # = 2 everywhere
    assert g[0] == 2.0 and g[N - 1] == 2.0 and ek.hmin(g).numpy()[0] == 2.0 and ek.hmax(g).numpy()[0] == 2.0
    assert ek.count(g == ek.Float32(2.0)) == N
    ek.hip_malloc_trim()
