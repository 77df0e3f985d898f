"""gather variants (elements per lane x table cache policy x table size) -- csrc/probe.hip k_probe_gather"""
import ctypes, os, statistics, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enoki_amd import capi, hiprt
capi.init(); st = capi.stream(); P = ctypes.c_void_p
n = 1 << 26
rng = np.random.default_rng(0)
out = capi.Buf(np.float32, n)
for logk in (16, 20, 22, 24):
    K = 1 << logk
    table = capi.Buf.from_numpy(rng.uniform(-1, 1, K).astype(np.float32))
    idx = capi.Buf.from_numpy(rng.integers(0, K, n).astype(np.uint32))
    fns = {(e, p): (lambda e=e, p=p: capi.check(capi.probe_lib().ek_hip_probe_gather(e, p, P(out.ptr), P(table.ptr), P(idx.ptr), ctypes.c_size_t(n))))
           for e in (1, 4, 8) for p in (0, 1)}
    fns[("prod", 0)] = lambda: capi.gather(table, idx)
    samples = {k: [] for k in fns}
    for _ in range(5):
        for k, f in fns.items():
            samples[k].append(hiprt.time_region(st, f, iters=10, warmup=2))
    for k, v in samples.items():
        ms = statistics.median(v)
        print(f"K=2^{logk} elems/lane={k[0]} table_policy={'nt' if k[1] else 'plain'}  {ms:7.4f} ms  {n / ms / 1e6:7.1f} G lookups/s  {12 * n / ms / 1e9:6.3f} TB/s algorithmic")
