"""cfg5 (synthetic, not in the reference): the 3-bounce path tracer of bench.py exercises PCG32, the second-wave
functions, gather and the scatter_add adjoint together.  There is no reference output, so the checks are internal
consistency: determinism, energy bounds, and the texture gradient against central finite differences (the loss is a
cubic polynomial in the texels, geometry does not depend on them)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_path_tracer_gradient_matches_finite_differences():
    import enoki_amd.hip as ekc
    import enoki_amd.hip_autodiff as ek
    from bench import path_trace
    n, width = 1 << 16, 64
    K = width * width
    rng = np.random.default_rng(1)
    tex_np = rng.uniform(0.2, 0.8, K).astype(np.float32)

    def loss_of(t, grad=False):
        tex = ek.Float32(ekc.Float32(t))
        if grad:
            ek.set_requires_gradient(tex)
        y = path_trace(ek, ekc, tex, n, seed=42, bounces=3, width=width)
        if not grad:
            return float(ek.detach(y).numpy()[0])
        ek.backward(y)
        return float(ek.detach(y).numpy()[0]), ek.gradient(tex).numpy()

    y0, g = loss_of(tex_np, grad=True)
    y1, g1 = loss_of(tex_np, grad=True)
    assert y0 == y1 and np.array_equal(g.view(np.uint32) if False else g, g1) or np.allclose(g, g1, rtol=1e-5)   # same paths, same loss
    # every path carries at most 0.1 (a + a^2 + a^3) + a^3 <= ~0.71 with a < 0.8 and at least 0.2^3
    assert 0.2 ** 3 * n < y0 < 0.72 * n
    assert g.shape == (K,) and np.all(g >= 0) and g.sum() > 0
    # directional derivative along a random direction
    v = rng.standard_normal(K).astype(np.float32)
    eps = np.float32(2e-2)
    fd = (loss_of(tex_np + eps * v) - loss_of(tex_np - eps * v)) / (2 * float(eps))
    an = float(np.dot(g.astype(np.float64), v.astype(np.float64)))
    assert abs(fd - an) <= 2e-2 * max(abs(an), 1.0), (fd, an)
    # texels that no path visits receive no gradient; most texels of a 64 x 64 texture are visited by 65536 x 3 hits
    assert (g > 0).mean() > 0.9


def test_path_tracer_matches_the_reference_build():
    """BASELINE configs[4] against its oracle: examples/path_trace.h instantiated on the reference's own arrays
    (oracle/ref_driver.cpp:ref_cfg5, unmodified reference headers) and the same program spelled with the python bindings visit the
    same texels (only class A operations decide where a path goes), so loss and texture gradient differ by the order of fp
    additions only: class D."""
    import enoki_amd.hip as ekc
    import enoki_amd.hip_autodiff as ek
    import oracle_lib as ol
    from bench import path_trace
    from conftest import stat_sum_bound, hsum_depth
    try:
        ref = ol.ref()
    except Exception:
        pytest.skip("oracle/_ref is not built")
    n, width = 1 << 20, 1024
    K = width * width
    tex_np = (np.random.default_rng(3).uniform(0.2, 0.8, K)).astype(np.float32)
    ry, rg, _ = ref.cfg5(tex_np, n, seed=42, first_lane=7, bounces=3, width=width)
    tex = ek.Float32(ekc.Float32(tex_np))
    ek.set_requires_gradient(tex)
    y = path_trace(ek, ekc, tex, n, seed=42, first_lane=7, bounces=3, width=width)
    ek.backward(y)
    yv, g = float(ek.detach(y).numpy()[0]), ek.gradient(tex).numpy()
    eps = 2.0 ** -24
    # every path carries between 0.2^3 and 0.72: the loss is a sum of n positive terms
    assert abs(yv - ry) <= eps * (hsum_depth(n) + n // 8 + 8) * max(abs(ry), abs(yv)), (yv, ry)
    # per texel: a handful of positive terms each (3 n hits over K texels); the same texels are hit on both sides
    assert np.array_equal(g == 0, rg == 0)
    hits = np.maximum(np.ceil(np.maximum(g, rg) / (0.1 * 0.2 * 0.2)), 1.0)          # no term is smaller than 0.1 * 0.2^2
    assert np.all(np.abs(g - rg) <= eps * (hits + 4) * np.maximum(g, rg)), float(np.abs(g - rg).max())


@pytest.mark.parametrize("entry", ["path_trace_fused_device", "path_trace_device"])
def test_cpp_path_tracers_match_the_reference_build(entry):
    """examples/libpath_trace.so: the template of examples/path_trace.h (a) on one-element packets inside ONE kernel, with
    forward-mode duals for the three texture lookups, one tape node and one scatter_add in backward(); (b) on
    DiffArray<HIPArray<float>> op by op -- both against the same template on the reference's arrays (ref_cfg5)."""
    import ctypes
    import enoki_amd.hip as ekc
    import oracle_lib as ol
    from conftest import hsum_depth
    try:
        ref = ol.ref()
    except Exception:
        pytest.skip("oracle/_ref is not built")
    ekc.hip_init(0)
    lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "libpath_trace.so"))
    n, width = 1 << 20, 1024
    K = width * width
    tex_np = (np.random.default_rng(3).uniform(0.2, 0.8, K)).astype(np.float32)
    ry, rg, _ = ref.cfg5(tex_np, n, seed=42, first_lane=7, bounces=3, width=width)
    tex, loss, grad = ekc.Float32(tex_np), ekc.Float32.empty(1), ekc.Float32.empty(K)
    P = ctypes.c_void_p
    rc = getattr(lib, entry)(P(tex.data_ptr()), ctypes.c_size_t(K), ctypes.c_size_t(n), ctypes.c_uint64(42), ctypes.c_uint64(7), 3,
                             ctypes.c_uint32(width), P(loss.data_ptr()), P(grad.data_ptr()))
    assert rc == 0
    yv, g = float(loss.numpy()[0]), grad.numpy()
    eps = 2.0 ** -24
    assert abs(yv - ry) <= eps * (hsum_depth(n) + n // 8 + 8) * max(abs(ry), abs(yv)), (yv, ry)
    assert np.array_equal(g == 0, rg == 0)                                    # the same texels are hit
    hits = np.maximum(np.ceil(np.maximum(g, rg) / (0.1 * 0.2 * 0.2)), 1.0)
    # (the dual numbers of the fused kernel form each derivative in another association than the tape's products: a few ulp per term)
    assert np.all(np.abs(g - rg) <= eps * (4 * hits + 8) * np.maximum(g, rg)), float((np.abs(g - rg) / np.maximum(g, 1e-30)).max())


def test_fused_path_tracer_at_the_bench_size_matches_the_reference_build():
    """The size bench.py quotes (16 Mi paths per GPU, its seed, K = 1 Mi): loss and texture gradient of the ONE fused kernel against
    examples/path_trace.h instantiated on the reference's arrays (ref_cfg5: ~12 s and ~6 GB of host memory).  Same bounds as at
    1 Mi paths; per texel ~48 hits instead of 3."""
    import ctypes
    import enoki_amd.hip as ekc
    import oracle_lib as ol
    from conftest import hsum_depth
    try:
        ref = ol.ref()
    except Exception:
        pytest.skip("oracle/_ref is not built")
    ekc.hip_init(0)
    from enoki_amd import synth
    import bench
    lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "libpath_trace.so"))
    n, width, K, seed = bench.N_PATHS_PER_GPU, 1024, bench.K_TABLE, 0x853c49e6748fea9b
    assert n == 1 << 24 and K == width * width
    tex = ekc.fmadd(synth.uniform_pm1(0, K, 8), ekc.Float32(0.3), ekc.Float32(0.5))        # bench.py's texture: albedo in [0.2, 0.8)
    tex_np = tex.numpy()
    ry, rg, _ = ref.cfg5(tex_np, n, seed=seed, first_lane=0, bounces=3, width=width)
    loss, grad = ekc.Float32.empty(1), ekc.Float32.empty(K)
    P = ctypes.c_void_p
    rc = lib.path_trace_fused_device(P(tex.data_ptr()), ctypes.c_size_t(K), ctypes.c_size_t(n), ctypes.c_uint64(seed), ctypes.c_uint64(0), 3,
                                     ctypes.c_uint32(width), P(loss.data_ptr()), P(grad.data_ptr()))
    assert rc == 0
    yv, g = float(loss.numpy()[0]), grad.numpy()
    eps = 2.0 ** -24
    assert abs(yv - ry) <= eps * (hsum_depth(n) + n // 8 + 8) * max(abs(ry), abs(yv)), (yv, ry)
    assert np.array_equal(g == 0, rg == 0)                                    # the same texels are hit
    hits = np.maximum(np.ceil(np.maximum(g, rg) / (0.1 * 0.2 * 0.2)), 1.0)
    assert np.all(np.abs(g - rg) <= eps * (4 * hits + 8) * np.maximum(g, rg)), float((np.abs(g - rg) / np.maximum(g, 1e-30)).max())
