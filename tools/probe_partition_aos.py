"""64-way partition of (idx, c, x) into 12-byte bucket-ordered records, one store per element, space reserved per tile
with global atomics (csrc/probe.hip k_probe_partition_aos) -- against the product's SoA pipeline (count + scan +
partition).  GPU box: python tools/probe_partition_aos.py"""
import ctypes, os, statistics, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enoki_amd import capi, hiprt
capi.init(); st = capi.stream(); P = ctypes.c_void_p
pl = capi.probe_lib()
n, K = 1 << 26, 1 << 20
rng = np.random.default_rng(0)
idx_h = rng.integers(0, K, n).astype(np.uint32)
idx = capi.Buf.from_numpy(idx_h)
c = capi.Buf.from_numpy(rng.uniform(-1, 1, n).astype(np.float32))
x = capi.Buf.from_numpy(rng.uniform(-1, 1, n).astype(np.float32))
cap = int(n / 64 * 1.02)
out = capi.Buf(np.uint32, 64 * cap * 3)
cursor = capi.Buf(np.uint32, 64)
names = {0: "4096-record tiles, 512 threads, 3 wg/CU", 1: "4096-record tiles, 1024 threads", 2: "8192-record tiles, 512 threads, 1 wg/CU",
         3: "2048-record tiles, 256 threads, 6 wg/CU", 4: "8192-record tiles, 1024 threads, 1 wg/CU"}
print(f"# tools/probe_partition_aos.py on 1 x MI355X: {n >> 20} Mi elements, K = {K >> 20} Mi bins, 64 buckets; 24 B/elt")
for v in range(5):
    f = lambda v=v: capi.check(pl.ek_hip_probe_partition_aos(v, P(out.ptr), P(cursor.ptr), ctypes.c_uint32(cap), P(idx.ptr), P(c.ptr), P(x.ptr), ctypes.c_size_t(n)))
    ms = statistics.median(hiprt.time_region(st, f, iters=10, warmup=2) for _ in range(5))
    f(); capi.sync()
    cur = cursor.numpy()
    ok = int(cur.sum()) <= n and int(cur.max()) <= cap
    # spot check bucket 3: the multiset of local indices
    rec = out.numpy().reshape(64, cap, 3)
    b = 3
    got = np.sort(rec[b, :cur[b], 0])
    sel = idx_h[: int(cur.sum())]
    exp = np.sort(sel[(sel >> 14) == b] & 0x3fff)
    print(f"{names[v]:48s} {ms:7.4f} ms  {n * 24 / ms / 1e9:6.3f} TB/s  records {int(cur.sum())}  bucket check {ok and np.array_equal(got, exp)}")
# the product pipeline for comparison
ta, tb = capi.Buf.from_numpy(np.zeros(K, np.float32)), capi.Buf.from_numpy(np.zeros(K, np.float32))
g = lambda: capi.scatter_add_multi([ta, tb], [c, c], idx, weights=[x, None])
g()
capi.profile_begin()
for _ in range(5): g()
for k in capi.profile_end(): print("product", k["kernel"], round(k["total_ms"] / k["launches"], 4), "ms")
