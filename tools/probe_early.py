"""The forward + adjoint kernel of the headline step alone (k_bucket_pair_forward_adjoint), on a partition that is made once:
per-launch time by HIP events on the library stream.  For A/B runs of kernel variants (build/variants/libenoki-hip-<name>.so
swapped over enoki_amd/libenoki-hip.so inside one gpurun call).  python tools/probe_early.py [log2 n] [log2 K] [label]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enoki_amd import capi, hiprt
capi.init(); st = capi.stream()
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 26
logk = int(sys.argv[2]) if len(sys.argv) > 2 else 20
label = sys.argv[3] if len(sys.argv) > 3 else ""
n, K = 1 << logn, 1 << logk
# PROBE_HINTS=1: the adjoint hint alone (half-size buckets, exchange locks); default 3: + bounded (quarter-size buckets, fixed point)
HINTS = int(os.environ.get("PROBE_HINTS", capi.Bucketed.HINT_ADJOINT | capi.Bucketed.HINT_BOUNDED))
rng = np.random.default_rng(0)
A = capi.Buf.from_numpy(rng.uniform(-1, 1, K).astype(np.float32))
B = capi.Buf.from_numpy(rng.uniform(-1, 1, K).astype(np.float32))
x = capi.Buf.from_numpy(rng.uniform(-1, 1, n).astype(np.float32))
idx = capi.Buf.from_numpy(rng.integers(0, K, n).astype(np.uint32))


def step():
    b = capi.Bucketed("fmadd", A, x, B, idx, hints=HINTS)
    y = b.reduce("hsum", "sin", keep=True, keep_op="cos")
    gA, gB = capi.fill(np.float32, 0, K), capi.fill(np.float32, 0, K)
    b.scatter_add([gB, gA], [("cos", 0, False), ("cos", 0, True)])
    b.destroy()
    return y


for _ in range(3):
    step()
capi.sync()
ms = min(hiprt.time_region(st, step, iters=10, warmup=1) for _ in range(3))
capi.profile_begin()
for _ in range(10):
    step()
rows = {k["kernel"]: k["total_ms"] / k["launches"] for k in capi.profile_end() if k["launches"]}
print(f"{label:10s} step {ms:7.4f} ms  " + "  ".join(f"{k} {v * 1e3:6.1f} us" for k, v in sorted(rows.items(), key=lambda kv: -kv[1])[:4]))
