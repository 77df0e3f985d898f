"""Gathers consumed in place (ek_hip_map_gathered, csrc/gathered.hip) and the deferred gathers of HIPArray that feed
them: bit-identical to ek_hip_gather followed by the plain vertical op (class A)."""
import numpy as np
import pytest

from conftest import bits_equal, f32_inputs, f64_inputs

pytestmark = pytest.mark.gpu

SIZES = [1, 3, 4, 5, 255, 256, 257, 4099, 100003, (1 << 20) + 7]


def up(capi, a):
    return capi.Buf.from_numpy(a)


def _case(dtype, n, K, seed, masked):
    rng = np.random.default_rng(seed)
    gen = f32_inputs if dtype == np.float32 else f64_inputs
    A = gen(K, seed=seed + 1).astype(dtype)
    B = gen(K, seed=seed + 2).astype(dtype)
    x = gen(n, seed=seed + 3).astype(dtype)
    y = gen(n, seed=seed + 4).astype(dtype)
    idx = rng.integers(0, K, n).astype(np.uint32)
    mask = (rng.integers(0, 4, n) != 0).astype(np.uint8) if masked else None
    return A, B, x, y, idx, mask


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("op", ["add", "sub", "mul"])
def test_binary_consumes_gather(capi, oracle, dtype, masked, op):
    for n in SIZES:
        A, B, x, y, idx, mask = _case(dtype, n, 1000, seed=n, masked=masked)
        dA, dx, di = up(capi, A), up(capi, x), up(capi, idx)
        dm = up(capi, mask) if masked else True
        ga = capi.gather(dA, di, dm)                      # plain gather kernel: the reference result
        for slot in (0, 1):
            ops = [capi.G(dA, di, dm), dx] if slot == 0 else [dx, capi.G(dA, di, dm)]
            ref = capi.binary(op, ga, dx) if slot == 0 else capi.binary(op, dx, ga)
            got = capi.map_gathered(op, *ops)
            assert bits_equal(got.numpy(), ref.numpy()), (op, n, slot)
        # scalar partner
        got = capi.map_gathered(op, capi.G(dA, di, dm), 1.5)
        assert bits_equal(got.numpy(), capi.binary(op, ga, 1.5).numpy()), (op, n)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("op", ["fmadd", "fmsub", "fnmadd", "fnmsub"])
def test_fma_consumes_gather(capi, oracle, dtype, masked, op):
    for n in SIZES:
        A, B, x, y, idx, mask = _case(dtype, n, 777, seed=n + 5, masked=masked)
        dA, dB, dx, dy, di = up(capi, A), up(capi, B), up(capi, x), up(capi, y), up(capi, idx)
        dm = up(capi, mask) if masked else True
        ga, gb = capi.gather(dA, di, dm), capi.gather(dB, di, dm)
        GA, GB = capi.G(dA, di, dm), capi.G(dB, di, dm)
        cases = {
            "A": ([GA, dx, dy], [ga, dx, dy]),
            "B": ([dx, GA, dy], [dx, ga, dy]),
            "C": ([dx, dy, GB], [dx, dy, gb]),
            "pair": ([GA, dx, GB], [ga, dx, gb]),
            "pair-swapped": ([dx, GA, GB], [dx, ga, gb]),
            "A scalar": ([GA, 0.75, dy], [ga, 0.75, dy]),
            "pair scalar": ([GA, -2.0, GB], [ga, -2.0, gb]),
        }
        for name, (fused, plain) in cases.items():
            got = capi.map_gathered(op, *fused, n=n)
            ref = capi.ternary(op, *plain, n=n)
            assert bits_equal(got.numpy(), ref.numpy()), (op, n, name)


def test_unsupported_combinations_are_reported(capi):
    A = up(capi, np.arange(100, dtype=np.float32)); B = up(capi, np.arange(50, dtype=np.float32))
    i1 = up(capi, np.zeros(4096, np.uint32)); i2 = up(capi, np.ones(4096, np.uint32))
    x = up(capi, np.ones(4096, np.float32))
    with pytest.raises(capi.EnokiHipError):      # pair through different index arrays
        capi.map_gathered("fmadd", capi.G(A, i1), x, capi.G(A, i2))
    with pytest.raises(capi.EnokiHipError):      # pair with tables of different size
        capi.map_gathered("fmadd", capi.G(A, i1), x, capi.G(B, i1))
    with pytest.raises(capi.EnokiHipError):      # both factors gathered
        capi.map_gathered("fmadd", capi.G(A, i1), capi.G(A, i1), x)
    with pytest.raises(capi.EnokiHipError):      # op that cannot consume a gather
        capi.map_gathered("div", capi.G(A, i1), x)


def test_misaligned_operands(capi):
    n = 10007
    A, B, x, y, idx, _ = _case(np.float32, n + 3, 513, seed=3, masked=False)
    dA, dB = up(capi, A[:513]), up(capi, B[:513])
    bx, bi = up(capi, x), up(capi, idx)             # the views below do not own their memory
    for off in (1, 2, 3):
        dx, di = bx.view(off, n), bi.view(off, n)
        got = capi.map_gathered("fmadd", capi.G(dA, di), dx, capi.G(dB, di), n=n)
        ref = capi.ternary("fmadd", capi.gather(dA, di), dx, capi.gather(dB, di), n=n)
        assert bits_equal(got.numpy(), ref.numpy())
        assert np.array_equal(capi.gather(dA, di).numpy().view(np.uint32), A[:513][idx[off:off + n]].view(np.uint32))


# ----------------------------------------------------------------------------------------------
#  HIPArray level: deferred gathers keep value semantics
# ----------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ek():
    import enoki_amd.hip as m
    m.hip_init(0)
    return m


def test_deferred_gather_semantics(ek):
    rng = np.random.default_rng(5)
    n, K = 50000, 1024
    A = rng.standard_normal(K).astype(np.float32); B = rng.standard_normal(K).astype(np.float32)
    x = rng.standard_normal(n).astype(np.float32); idx = rng.integers(0, K, n).astype(np.uint32)
    dA, dB, dx, di = ek.Float32(A), ek.Float32(B), ek.Float32(x), ek.UInt32(idx)
    ref = (A[idx].astype(np.float64) * x + B[idx]).astype(np.float32)

    l0 = ek.hip_launch_count()
    u = ek.fmadd(ek.gather(dA, di), dx, ek.gather(dB, di))
    assert ek.hip_launch_count() - l0 == 2            # interleave + ONE fused kernel, no gather launches
    assert bits_equal(u.numpy(), ref)

    # a deferred gather that is looked at directly is an ordinary array
    g = ek.gather(dA, di)
    assert bits_equal(g.numpy(), A[idx])
    # consumed twice: fused the first time, materialised for the second consumer; same values
    g = ek.gather(dA, di)
    p, q = g * dx, g + dx
    assert bits_equal(p.numpy(), A[idx] * x) and bits_equal(q.numpy(), A[idx] + x)
    # writing into the table AFTER the gather must not change the gathered values
    T = ek.Float32(A)
    g = ek.gather(T, di)
    ek.scatter(T, ek.Float32(np.full(K, 7.0, np.float32)), ek.UInt32.arange(K))
    assert bits_equal((g * dx).numpy(), A[idx] * x)
    assert np.all(T.numpy() == 7.0)
    # masked
    m = rng.integers(0, 2, n).astype(bool)
    g = ek.gather(dA, di, ek.Mask(m.astype(np.uint8)))
    assert bits_equal((g * dx).numpy(), np.where(m, A[idx], np.float32(0)) * x)
    # switched off: same results through the plain kernels
    ek.hip_set_defer_gather(False)
    try:
        l0 = ek.hip_launch_count()
        u2 = ek.fmadd(ek.gather(dA, di), dx, ek.gather(dB, di))
        assert ek.hip_launch_count() - l0 == 3
        assert bits_equal(u2.numpy(), ref)
    finally:
        ek.hip_set_defer_gather(True)
