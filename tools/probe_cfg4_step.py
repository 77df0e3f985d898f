"""cfg4_bucketed step anatomy: which part of the 0.38 ms is kernels, which is the host round trip (tools/, GPU box)"""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import enoki_amd.hip as ek
ek.hip_init(0)
nr = 1 << 25
res = 4096
F = ek.Float32
grid = ek.meshgrid(F.linspace(-1.2, 1.2, res), F.linspace(-1.2, 1.2, nr // res))
rng = np.random.default_rng(1)
perm = ek.UInt32(rng.permutation(nr).astype(np.uint32))
mask = ek.UInt32(rng.integers(0, 4, nr).astype(np.uint32)) != ek.UInt32(0)
lib = ctypes.CDLL(os.path.join(ROOT, "examples", "libsphere_fused.so"))
P = ctypes.c_void_p
hits = ctypes.c_uint64()
keep = {}


def through(image):
    rc = lib.sphere_through_device(P(grid.x.data_ptr()), P(grid.y.data_ptr()), P(perm.data_ptr()), P(mask.data_ptr()),
                                   ctypes.c_size_t(nr), P(image.data_ptr()), ctypes.byref(hits))
    assert rc == 0


def step_full():
    image = F.full(-1.0, nr)
    through(image)
    keep["image"] = image


fixed = F.full(-1.0, nr)
def step_reuse():
    through(fixed)


def step_two_fills():
    keep["a"] = F.full(-1.0, nr)
    image = F.full(-1.0, nr)
    through(image)
    keep["image"] = image


def step_fill():
    image = F.empty(nr)
    rc = lib.sphere_through_fill_device(P(grid.x.data_ptr()), P(grid.y.data_ptr()), P(perm.data_ptr()), P(mask.data_ptr()),
                                        ctypes.c_size_t(nr), ctypes.c_float(-1.0), P(image.data_ptr()), ctypes.byref(hits))
    assert rc == 0
    keep["image"] = image


def wall(label, fn, reps=20):
    for _ in range(3):
        fn()
    ek.hip_sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ek.hip_sync()
    dt = (time.perf_counter() - t0) / reps
    ek.hip_profile_begin()
    for _ in range(5):
        fn()
    p = json.loads(ek.hip_profile_end())
    parts = "  ".join(f"{k['kernel']} {k['total_ms'] / k['launches'] * 1e3:.0f}" + (f" x{k['launches'] // 5}" if k['launches'] > 5 else "")
                      for k in sorted(p, key=lambda k: -k["total_ms"]) if k["launches"])
    print(f"{label:28s} {dt * 1e3:.3f} ms wall   [us] {parts}")


wall("fill + through", step_full)
wall("through on a fixed image", step_reuse)
wall("two fills + through", step_two_fills)
wall("through_fill", step_fill)
step_full(); h0 = hits.value; a = keep["image"].numpy().copy()
step_fill(); assert hits.value == h0 and np.array_equal(a.view(np.uint32), keep["image"].numpy().view(np.uint32)), "fill variant differs"
print("fill variant: same image, same count", h0)
