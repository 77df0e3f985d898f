"""C gathers through one index array: separate launches vs one ek_hip_gather_multi launch (64 Mi lookups per table).
ENOKI_HIP_GATHER_MULTI=always forces the fused launch (the default picks by table size, HIPArray::gather_multi_)."""
import os, sys
os.environ["ENOKI_HIP_GATHER_MULTI"] = "always"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enoki_amd import capi, hiprt, synth
import enoki_amd.hip as ek
capi.init(); st = capi.stream()
n = 1 << 26
for C in (2, 3):
    for logk in (14, 16, 17, 18, 19, 20, 22, 24, 25, 26):
        K = 1 << logk
        T = [synth.uniform_pm1(0, K, 6 + c) for c in range(C)]
        idx = synth.index_mod(0, n, 4, K)
        V = ek.Vector2f(*T) if C == 2 else ek.Vector3f(*T)
        sep = lambda: [ek.gather(t, idx) for t in T]
        multi = lambda: ek.gather(V, idx)
        t1 = min(hiprt.time_region(st, sep, iters=5, warmup=1) for _ in range(3))
        t2 = min(hiprt.time_region(st, multi, iters=5, warmup=1) for _ in range(3))
        print(f"C={C} K=2^{logk} ({K * 4 >> 10} KiB per table): separate {t1:.4f} ms, gather_multi {t2:.4f} ms  ratio {t2 / t1:.2f}", flush=True)
