"""Morton / Z-order codes (include/enoki/morton.h; reference include/enoki/morton.h): integer bit work, exact."""
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def spread(x, dim, most):
    out = np.zeros_like(x)
    for b in range(most):
        out |= ((x >> np.array(b, x.dtype)) & np.array(1, x.dtype)) << np.array(b * dim, x.dtype)
    return out


def test_host_scalars_match_the_definition():
    out = subprocess.run([os.path.join(HERE, "cpp", "morton_host.bin")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.extras
@pytest.mark.parametrize("dim", [2, 3])
def test_device_arrays(dim):
    import enoki_amd.hip as ek
    ek.hip_init(0)
    rng = np.random.default_rng(dim)
    n = 100003
    coords = [rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32) for _ in range(dim)]
    Vec = getattr(ek, f"Vector{dim}u")
    code = ek.morton_encode(Vec(*[ek.UInt32(c) for c in coords])).numpy()
    most = 32 // dim
    want = np.zeros(n, np.uint32)
    for i, c in enumerate(coords):
        want |= spread(c, dim, most) << np.uint32(i)
    assert np.array_equal(code, want)
    back = Vec.morton_decode(ek.UInt32(code))
    for i, c in enumerate(coords):
        assert np.array_equal(back[i].numpy(), c & np.uint32((1 << most) - 1))
