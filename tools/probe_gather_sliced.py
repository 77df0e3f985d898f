"""L2 blocking in time for gathers: S launches, each serving the lookups of one table slice (csrc/probe.hip k_probe_gather_range)"""
import ctypes, os, statistics, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enoki_amd import capi, hiprt
capi.init(); st = capi.stream(); P = ctypes.c_void_p
n = 1 << 26
rng = np.random.default_rng(0)
out = capi.Buf(np.float32, n)
for logk in (20, 22, 24):
    K = 1 << logk
    tab = rng.uniform(-1, 1, K).astype(np.float32)
    table = capi.Buf.from_numpy(tab)
    ih = rng.integers(0, K, n).astype(np.uint32)
    idx = capi.Buf.from_numpy(ih)
    base = lambda: capi.check(capi.probe_lib().ek_hip_probe_gather(4, 0, P(out.ptr), P(table.ptr), P(idx.ptr), ctypes.c_size_t(n)))
    ms = statistics.median(hiprt.time_region(st, base, iters=10, warmup=2) for _ in range(3))
    print(f"K=2^{logk} one launch            {ms:7.4f} ms", flush=True)
    for S in (1, 2, 4, 8):
        f = lambda S=S: capi.check(capi.probe_lib().ek_hip_probe_gather_sliced(S, P(out.ptr), P(table.ptr), ctypes.c_size_t(K), P(idx.ptr), ctypes.c_size_t(n)))
        f()
        ok = np.array_equal(out.numpy()[:100000], tab[ih[:100000]])
        ms = statistics.median(hiprt.time_region(st, f, iters=10, warmup=2) for _ in range(3))
        print(f"K=2^{logk} {S} slice launch(es)   {ms:7.4f} ms  correct={ok}", flush=True)
