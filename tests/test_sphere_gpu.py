"""BASELINE config 4: masked gather / scatter ray-sphere intersection (tests/sphere.cpp, tests/ray.h) on
Array<HIPArray<float>, 3>.  Every op on this path is class A (mul/add/fma/sqrt/div/select/max, gather, scatter with a
permutation, count), so the image must be BIT-EXACT against the reference build and the hit count equal."""
import ctypes
import os

import numpy as np
import pytest

import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))
HAVE_REF = os.path.exists(os.path.join(ol.ORACLE_DIR, "_ref", "libenoki_ref.so"))


def scene(res, seed=0, shard=None):
    n = res * res
    lin = ol.port().linspace(-1.2, 1.2, res) if False else None
    step = (np.float32(1.2) - np.float32(-1.2)) / np.float32(res - 1)
    lin = (np.arange(res, dtype=np.float32) * step + np.float32(-1.2)).astype(np.float32)   # fmadd(i, step, min) below
    lin = np.array([np.float32(np.float64(i) * np.float64(step) + np.float64(np.float32(-1.2))) for i in range(res)], np.float32)
    gx, gy = np.tile(lin, res), np.repeat(lin, res)
    rng = np.random.default_rng(seed)
    perm = rng.permutation(n).astype(np.uint32)
    mask = (rng.integers(0, 4, n) != 0).astype(np.uint8)
    return gx, gy, perm, mask


def run(fn, gx, gy, perm, mask):
    n = gx.size
    img = np.full(n, -1.0, np.float32)
    hc = ctypes.c_uint64()
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = fn(p(gx), p(gy), p(perm), p(mask), ctypes.c_size_t(n), p(img), ctypes.byref(hc))
    assert rc == 0
    return img, hc.value


def fill_entry(lib):
    """sphere_through_fill(): `image = full(-1)` happens inside the call, the image passed in is an output only -- hand it
    garbage to prove that"""
    def fn(gx, gy, perm, mask, n, img, hc):
        ctypes.memset(img, 0x5a, n.value * 4)
        return lib.sphere_through_fill(gx, gy, perm, mask, n, ctypes.c_float(-1.0), img, hc)
    return fn


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref was not built")
@pytest.mark.parametrize("res", [8, 64, 257])
def test_oracle_cfg4_matches_reference_build(res):
    args = scene(res)
    ri, rh = run(ol.ref().lib.ref_cfg4, *args)
    pi, ph = run(ol.port().lib.orc_cfg4, *args)
    assert rh == ph and np.array_equal(ri.view(np.uint32), pi.view(np.uint32))


def test_oracle_cfg4_matches_golden():
    z = np.load(os.path.join(HERE, "golden", "cfg4.npz"))
    pi, ph = run(ol.port().lib.orc_cfg4, z["gx"], z["gy"], z["perm"], z["mask"])
    assert ph == int(z["hits"]) and np.array_equal(pi.view(np.uint32), z["image"].view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("res", [8, 64, 257, 1024])
def test_hip_cfg4_bit_exact(res):
    lib = ctypes.CDLL(os.path.join(HERE, "cpp", "libsphere_hip.so"))
    args = scene(res, seed=res)
    gi, gh = run(lib.hip_cfg4, *args)
    pi, ph = run(ol.port().lib.orc_cfg4, *args)           # oracle (pinned to the reference build above)
    assert gh == ph
    assert np.array_equal(gi.view(np.uint32), pi.view(np.uint32))
    if HAVE_REF:
        ri, rh = run(ol.ref().lib.ref_cfg4, *args)
        assert rh == gh and np.array_equal(ri.view(np.uint32), gi.view(np.uint32))
    assert gh > 0 and gi.max() > 100


@pytest.mark.gpu
@pytest.mark.parametrize("res", [8, 64, 257, 1024])
@pytest.mark.parametrize("entry", ["sphere_fused", "sphere_fused_packed", "sphere_through", "sphere_through_fill"])
def test_hip_cfg4_fused_bit_exact(res, entry):
    """the same program as ONE kernel through enoki::vectorize() (examples/sphere_fused.cpp): bit-identical image, with
    the pixel grid as two planes or as packed {x, y} records (ONE 8-byte lookup per ray, array.h gather_packed); and
    executed per PIXEL, bucket by bucket (enoki::vectorize_through, enoki/vectorize_indexed.h: rays grouped by pixel bucket
    once, grid slices read and image slices written in order)"""
    lib = ctypes.CDLL(os.path.join(HERE, "..", "examples", "libsphere_fused.so"))
    args = scene(res, seed=res + 1)
    gi, gh = run(fill_entry(lib) if entry == "sphere_through_fill" else getattr(lib, entry), *args)
    pi, ph = run(ol.port().lib.orc_cfg4, *args)
    assert gh == ph and np.array_equal(gi.view(np.uint32), pi.view(np.uint32))
    assert gh > 0 and gi.max() > 100


def test_packed_record_gather_scatter_on_host_packets():
    """gather<Array<Packet, N>>(mem, index, mask) / scatter of packed records (array.h; array_router.h:1097-1107)"""
    import subprocess
    out = subprocess.run([os.path.join(HERE, "cpp", "packed_host.bin")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "packed_host:" in out.stdout


@pytest.mark.gpu
def test_hip_meshgrid_linspace_grid():
    """the pixel grid built by the product's own linspace + meshgrid (closed form fmadd(i, step, min))"""
    lib = ctypes.CDLL(os.path.join(HERE, "cpp", "libsphere_hip.so"))
    res = 300
    gx = np.zeros(res * res, np.float32); gy = np.zeros(res * res, np.float32)
    assert lib.hip_sphere_grid(ctypes.c_size_t(res), gx.ctypes.data_as(ctypes.c_void_p), gy.ctypes.data_as(ctypes.c_void_p)) == 0
    egx, egy, _, _ = scene(res)
    assert np.array_equal(gx.view(np.uint32), egx.view(np.uint32)) and np.array_equal(gy.view(np.uint32), egy.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("is_double", [0, 1])
def test_fused_math_matches_kernels(is_double):
    """every floating point function inside a vectorize() kernel (one-element packets -> the device algorithms of
    enoki/device/ek_math.h) returns the bits of the stand-alone kernels: 21 expressions, fused vs op by op, compared on
    the device (tests/cpp/vectorize_math_hip.cpp)"""
    lib = ctypes.CDLL(os.path.join(HERE, "cpp", "libvectorize_math_hip.so"))
    lib.vectorize_math_check.restype = ctypes.c_int
    report = ctypes.create_string_buffer(4096)
    for n in (1, 1000, (1 << 20) + 3):
        bad = lib.vectorize_math_check(is_double, ctypes.c_size_t(n), report, ctypes.c_size_t(len(report)))
        assert bad == 0, (n, report.value.decode())


@pytest.mark.gpu
def test_vectorize_through_with_duplicate_and_sparse_indices():
    """vectorize_through() does not need a permutation: an index array with duplicates and holes gives the image and the
    count(hit & mask) of the element-order program (every duplicate computes the same value; the count is per ELEMENT)"""
    lib = ctypes.CDLL(os.path.join(HERE, "..", "examples", "libsphere_fused.so"))
    for res, seed in ((64, 3), (300, 4), (1200, 5)):
        gx, gy, perm, mask = scene(res, seed=seed)
        n = perm.size
        rng = np.random.default_rng(seed)
        idx = rng.integers(0, n, n).astype(np.uint32)              # duplicates, and pixels nobody points at
        idx[: n // 7] = idx[n // 7: 2 * (n // 7)][: n // 7]
        fi, fh = run(lib.sphere_fused, gx, gy, idx, mask)          # the per-element fused kernel: same semantics
        for entry in (lib.sphere_through, fill_entry(lib)):
            gi, gh = run(entry, gx, gy, idx, mask)
            assert gh == fh and np.array_equal(gi.view(np.uint32), fi.view(np.uint32)), res
            none = np.zeros(n, np.uint8)                            # nothing active: the image is untouched, no hits
            gi, gh = run(entry, gx, gy, idx, none)
            assert gh == 0 and np.all(gi == -1.0)
        # more than 255 elements on one pixel: the byte counters of a bucket overflow and the call falls back to bitmaps
        for dup in (255, 256, 1000, n):
            heavy = idx.copy()
            heavy[:dup] = heavy[0] = np.uint32(n // 2 + res // 2)   # a pixel the sphere covers
            fi, fh = run(lib.sphere_fused, gx, gy, heavy, mask)
            for entry in (lib.sphere_through, fill_entry(lib)):
                gi, gh = run(entry, gx, gy, heavy, mask)
                assert gh == fh and np.array_equal(gi.view(np.uint32), fi.view(np.uint32)), (res, dup)
