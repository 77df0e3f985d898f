"""Real spherical harmonics (include/enoki/sh.h; reference include/enoki/sh.h: generated code for orders 0..9)."""
import ctypes
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_host_scalars_match_scipy_and_the_reference():
    from scipy.special import sph_harm_y
    lib = ctypes.CDLL(os.path.join(HERE, "cpp", "libsh_host.so"))
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rng = np.random.default_rng(1)
    n, order = 500, 12                                        # beyond the reference's limit of 9
    d = rng.standard_normal((3, n)); d = np.ascontiguousarray(d / np.linalg.norm(d, axis=0))
    out = np.empty(((order + 1) ** 2, n))
    lib.sh_host_f64(p(d), ctypes.c_size_t(n), ctypes.c_size_t(order), p(out))
    theta, phi = np.arccos(d[2]), np.arctan2(d[1], d[0])
    for l in range(order + 1):
        for m in range(-l, l + 1):
            Y = sph_harm_y(l, abs(m), theta, phi)             # complex, with the Condon-Shortley phase
            want = Y.real if m == 0 else np.sqrt(2) * (Y.real if m > 0 else Y.imag)
            assert np.abs(out[l * (l + 1) + m] - want).max() < 1e-12, (l, m)
    # the reference's generated code, float32, order 9 (tests/golden/sh.npz from oracle/_ref)
    z = np.load(os.path.join(HERE, "golden", "sh.npz"))
    mine = np.empty_like(z["out"])
    lib.sh_host_f32(p(z["d"]), ctypes.c_size_t(z["d"].shape[1]), ctypes.c_size_t(9), p(mine))
    assert np.abs(mine - z["out"]).max() < 4e-6


@pytest.mark.extras
def test_device_arrays_match_the_reference():
    import enoki_amd.hip as ek
    ek.hip_init(0)
    z = np.load(os.path.join(HERE, "golden", "sh.npz"))
    d = ek.Vector3f(*[ek.Float32(z["d"][i]) for i in range(3)])
    out = ek.sh_eval(d, 9)
    assert len(out) == 100
    n = z["d"].shape[1]
    for k in range(100):
        assert np.abs(np.broadcast_to(out[k].numpy(), (n,)) - z["out"][k]).max() < 4e-6, k
