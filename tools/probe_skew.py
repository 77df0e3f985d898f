import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from enoki_amd import capi, hiprt
capi.init(); st = capi.stream()
n = 1 << 24; K = 1 << 20
rng = np.random.default_rng(0)
vals = np.ones(n, np.float32)
for name, idx in (("all_zero", np.zeros(n, np.uint32)), ("two_bins", (np.arange(n) % 2).astype(np.uint32) * 70000),
                  ("zipf", np.minimum(rng.zipf(1.3, n) - 1, K - 1).astype(np.uint32)),
                  ("one_bucket", rng.integers(0, 16384, n).astype(np.uint32)), ("uniform", rng.integers(0, K, n).astype(np.uint32))):
    t = capi.fill(np.float32, 0.0, K); v = capi.Buf.from_numpy(vals); i = capi.Buf.from_numpy(idx)
    ms = hiprt.time_region(st, lambda: capi.scatter_add(t, v, i), iters=3, warmup=1)
    got = t.numpy() / 4.0      # 1 warmup + 3 iterations
    want = np.bincount(idx, minlength=K).astype(np.float64)
    ok = np.allclose(got, want, rtol=1e-6)
    print(f"{name:10s} {ms:8.3f} ms per call  {n / ms / 1e6:8.2f} G adds/s  exact counts: {ok}", flush=True)
