"""The reference's OWN test sources, compiled unmodified against this repository's headers with the device array types
substituted for the CPU ones (tests/cpp/refshim; built by enoki_amd/_build.py where /root/reference exists, the binary
travels to the GPU box).  SURVEY 8b(i): this is the templated code of the reference acting as the caller of the
HIPArray / DiffArray<HIPArray> member concept."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(name):
    exe = os.path.join(HERE, "cpp", name + ".bin")
    if not os.path.exists(exe):
        pytest.skip(f"{exe} was not built (needs /root/reference at build time)")
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    print(r.stdout)
    m = re.search(r"(\d+)/(\d+) passed", r.stdout)
    assert m, r.stdout[-2000:]
    return int(m.group(1)), int(m.group(2)), r.stdout


def test_reference_autodiff_suite_on_device():
    """tests/autodiff.cpp of the reference: 47 tests on DiffArray<HIPArray<float>>"""
    passed, total, log = _run("reftest_autodiff_hip")
    assert total == 47, log[-3000:]
    assert passed == total, "\n".join(l for l in log.splitlines() if "failure --" in l or "FAILED" in l)


def test_reference_sphere_program_on_device(tmp_path):
    """tests/sphere.cpp of the reference (ray.h's ENOKI_STRUCT Ray, make_rays / intersect_rays / shade_hits through
    vectorize()): compiled unmodified by hipcc, every vectorize() call is ONE fused kernel.  The program has no
    assertion of its own; its two images (separate kernels, combined kernel) must equal the CPU oracle's pixel for pixel."""
    import ctypes
    import numpy as np
    import oracle_lib as ol
    from test_sphere_gpu import run, scene
    exe = os.path.join(HERE, "cpp", "reftest_sphere_hip.bin")
    if not os.path.exists(exe):
        pytest.skip(f"{exe} was not built (needs /root/reference at build time)")
    r = subprocess.run([exe], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout
    # linspace + meshgrid (2 kernels for linspace/arange products, 2 gathers ...) and 3 + 1 vectorize kernels: the ray
    # tracing itself is 4 launches, not ~80
    launches = int(re.search(r"kernel launches: (\d+)", r.stdout).group(1))
    assert launches <= 16, r.stdout
    # the oracle on the same 1024 x 1024 grid, identity permutation, every ray active
    res = 1024
    gx, gy, _, _ = scene(res)
    perm = np.arange(res * res, dtype=np.uint32); mask = np.ones(res * res, np.uint8)
    img, hits = run(ol.port().lib.orc_cfg4, gx, gy, perm, mask)
    # write_image() prints (int) v; the oracle's cfg4 leaves missed pixels at their initial -1 where the reference program
    # shades the zero vector of a miss: 0.2 + max(dot(0, light), 0) * 90 = 0.2 -> 0
    expect = np.where(img < 0, 0, img.astype(np.int32))
    for name in ("sphere1.ppm", "sphere2.ppm"):
        tokens = open(os.path.join(tmp_path, name)).read().split()
        assert tokens[:4] == ["P3", "1024", "1024", "255"]
        got = np.array(tokens[4:], dtype=np.int64).reshape(-1, 3)
        bad = np.flatnonzero(got[:, 0] != expect)
        assert bad.size == 0, (name, bad.size, bad[:8], got[bad[:8], 0], expect[bad[:8]])
        assert np.array_equal(got[:, 1], expect) and np.array_equal(got[:, 2], expect), name
    assert hits > 0


def test_reference_headers_drive_the_backend():
    """integration/enoki/hip.h is written against the REFERENCE's headers: ArrayBase, array_router.h, array_math.h,
    array_struct.h dispatch into its member concept, which forwards to the C ABI.  tests/cpp/reference_side_hip.cpp
    instantiates the same templated functions on DynamicArray<Packet<float>> (reference CPU path) and on that HIPArray in
    one binary: class-A operations must agree bit for bit, reductions / rsqrt to their documented bounds."""
    exe = os.path.join(HERE, "cpp", "reference_side_hip.bin")
    if not os.path.exists(exe):
        pytest.skip("built only where /root/reference exists (enoki_amd/_build.py)")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1000:]
    assert "16/16 checks passed" in out.stdout, out.stdout[-2000:]


def test_reference_tape_and_tests_on_the_backend():
    """The reference's own tape (src/autodiff/autodiff.cpp), its own autodiff suite (tests/autodiff.cpp) and its own headers,
    all unmodified, with integration/enoki/hip.h + integration/hip_hooks.cpp as the array backend: 47 / 47 on the device.
    (tests/cpp/reftest_autodiff_hip.bin is the mirror image: the same suite on THIS repository's headers and tape.)"""
    exe = os.path.join(HERE, "cpp", "reference_tape_hip.bin")
    if not os.path.exists(exe):
        pytest.skip("built only where /root/reference exists (enoki_amd/_build.py)")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1000:]
    assert "47/47 passed" in out.stdout, out.stdout[-2000:]


def test_programs_written_with_the_reference_names_compile_unchanged():
    """<enoki/cuda.h> / CUDAArray / cuda_eval() and <enoki/dynamic.h> / DynamicArray<Packet<T>> are source-compatible aliases
    (include/enoki/cuda.h, include/enoki/dynamic.h): tests/cpp/compat_names_hip.cpp differentiates through a gather with
    them and checks the gradient analytically"""
    exe = os.path.join(HERE, "cpp", "compat_names_hip.bin")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr

