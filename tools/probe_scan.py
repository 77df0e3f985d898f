"""psum (single-pass decoupled look-back, csrc/scan.hip) and partition() timings.  GPU box: python tools/probe_scan.py"""
import ctypes, os, statistics, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enoki_amd import capi, hiprt
capi.init(); st = capi.stream()
print("# tools/probe_scan.py on 1 x MI355X: inclusive prefix sum, one pass over the data (4 B read + 4 B written per 4-byte element)")
for dt in (np.uint32, np.float32, np.float64):
    for logn in (20, 24, 26):
        n = 1 << logn
        a = capi.Buf.from_numpy(np.ones(n, dt))
        ms = statistics.median(hiprt.time_region(st, lambda: capi.psum(a), iters=10, warmup=2) for _ in range(5))
        b = 2 * n * np.dtype(dt).itemsize
        print(f"psum {np.dtype(dt).name:8s} n=2^{logn}  {ms:7.4f} ms  {b / ms / 1e9:6.3f} TB/s algorithmic ({b / ms / 1e9 / 8.0 * 100:4.1f} % of 8 TB/s)")
    if dt is np.float32:
        capi.set_tuning("deterministic", 1)
        n = 1 << 26
        a = capi.Buf.from_numpy(np.ones(n, dt))
        ms = statistics.median(hiprt.time_region(st, lambda: capi.psum(a), iters=10, warmup=2) for _ in range(5))
        print(f"psum float32  n=2^26, deterministic (three fixed-shape passes)  {ms:7.4f} ms")
        capi.set_tuning("deterministic", 0)
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "cpp", "libcall_hip.so"))
print("# partition() of 2^20 / 2^24 lanes (sort-based; cold, includes the host read-back of the run table)")
for n in (1 << 20, 1 << 24):
    for instances in (1, 3, 32, 1000, 70000):
        rng = np.random.default_rng(instances)
        which = rng.integers(0, instances, n).astype(np.uint32)
        gi = np.zeros(instances + 1, np.uint32); gs = np.zeros(instances + 1, np.uint32); ng = ctypes.c_uint32()
        perm = np.zeros(n, np.uint32); ms = ctypes.c_double(); launches = ctypes.c_uint64()
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        best = 1e9
        for _ in range(3):
            assert lib.hip_partition_many(p(which), ctypes.c_size_t(n), ctypes.c_uint32(instances), p(gi), p(gs), ctypes.byref(ng), p(perm),
                                          ctypes.byref(ms), ctypes.byref(launches)) == 0
            best = min(best, ms.value)
        print(f"partition n=2^{int(np.log2(n))} instances={instances:6d}  {best:8.3f} ms  {launches.value} launches")
