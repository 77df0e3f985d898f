import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from enoki_amd import capi, hiprt
capi.init(); st = capi.stream()
n = 1 << 22; K = 1 << 20
rng = np.random.default_rng(0)
vals = rng.uniform(-1, 1, n).astype(np.float32)
for name, idx in (("all_zero", np.zeros(n, np.uint32)), ("zipf", np.minimum(rng.zipf(1.3, n) - 1, K - 1).astype(np.uint32)),
                  ("uniform", rng.integers(0, K, n).astype(np.uint32))):
    t = capi.fill(np.float32, 0.0, K); v = capi.Buf.from_numpy(vals); i = capi.Buf.from_numpy(idx)
    ms = hiprt.time_region(st, lambda: capi.scatter_add(t, v, i, mode=1), iters=1, warmup=1)
    print(f"deterministic {name:10s} {ms:9.3f} ms per call  {n / ms / 1e6:8.3f} G adds/s", flush=True)
for name, idx in (("all_zero", np.zeros(n, np.uint32)), ("zipf", np.minimum(rng.zipf(1.3, n) - 1, K - 1).astype(np.uint32))):
    t = capi.fill(np.float32, 0.0, K); v = capi.Buf.from_numpy(vals); i = capi.Buf.from_numpy(idx)
    capi.profile_begin()
    capi.scatter_add(t, v, i, mode=1)
    for k in capi.profile_end():
        print(name, k["kernel"], k["launches"], round(k["total_ms"], 3))
