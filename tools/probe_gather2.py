"""gather table-load variants at 4 elements per lane: 0 plain global load, 1 nt, 2 agent-scope (sc1), 3 raw buffer load,
4 buffer load sc0, 5 buffer load sc0 sc1 -- csrc/probe.hip k_probe_gather"""
import ctypes, os, statistics, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enoki_amd import capi, hiprt
capi.init(); st = capi.stream(); P = ctypes.c_void_p
n = 1 << 26
rng = np.random.default_rng(0)
out = capi.Buf(np.float32, n)
ref = None
for logk in (16, 20, 24):
    K = 1 << logk
    tab = rng.uniform(-1, 1, K).astype(np.float32)
    table = capi.Buf.from_numpy(tab)
    ih = rng.integers(0, K, n).astype(np.uint32)
    idx = capi.Buf.from_numpy(ih)
    for p in (0, 1, 2, 3, 4, 5):
        f = lambda p=p: capi.check(capi.probe_lib().ek_hip_probe_gather(4, p, P(out.ptr), P(table.ptr), P(idx.ptr), ctypes.c_size_t(n)))
        f()
        ok = np.array_equal(out.numpy()[:100000], tab[ih[:100000]])
        ms = statistics.median(hiprt.time_region(st, f, iters=10, warmup=2) for _ in range(5))
        print(f"K=2^{logk} policy={p}  {ms:7.4f} ms  {n / ms / 1e6:7.1f} G lookups/s  correct={ok}", flush=True)
