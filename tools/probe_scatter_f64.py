"""float64 / int64 scatter_add (global atomics) next to the LDS-binned float32 path: 64 Mi adds into K bins"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enoki_amd import capi, hiprt, synth
import enoki_amd.hip as ek
capi.init(); st = capi.stream()
n = 1 << 26
v32 = synth.uniform_pm1(0, n, 3); v64 = ek.Float64(v32)
for logk in (14, 20, 24):
    K = 1 << logk
    idx = synth.index_mod(0, n, 4, K)
    for name, vals, zero in (("f32", v32, ek.Float32.zero), ("f64", v64, ek.Float64.zero)):
        t = zero(K)
        f = lambda: ek.scatter_add(t, vals, idx)
        ms = min(hiprt.time_region(st, f, iters=3, warmup=1) for _ in range(2))
        print(f"K=2^{logk} {name}: {ms:8.3f} ms  {n / ms / 1e6:7.1f} G adds/s", flush=True)
