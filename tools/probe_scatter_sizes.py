"""scatter_add throughput by table size (uniform indices, 64 Mi f32 adds).  GPU box: python tools/probe_scatter_sizes.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enoki_amd import capi, hiprt, synth  # noqa: E402
import enoki_amd.hip as ek  # noqa: E402

capi.init(); st = capi.stream()
n = 1 << 26
vals = synth.uniform_pm1(0, n, 3)
for logk in (8, 10, 14, 16, 18, 20, 22, 24, 26):
    K = 1 << logk
    idx = synth.index_mod(0, n, 4, K)
    t = ek.Float32.zero(K)
    f = lambda: ek.scatter_add(t, vals, idx)
    ms = min(hiprt.time_region(st, f, iters=5, warmup=1) for _ in range(2))
    total = float(ek.hsum(t).numpy()[0])
    print(f"K=2^{logk:2d}  {ms:8.3f} ms  {n / ms / 1e6:7.1f} G adds/s", flush=True)
