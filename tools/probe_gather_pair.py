"""Shared-index gather pair a = gather(A, idx), b = gather(B, idx) and its consumer u = fmadd(a, x, b):
separate 4-byte lookups vs ONE 8-byte lookup from an interleaved table, materialised vs consumed in place
(csrc/probe.hip k_probe_gather_pair).  GPU box: python tools/probe_gather_pair.py > gpurun_out/probe_gather_pair.txt"""
import ctypes, os, statistics, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enoki_amd import capi, hiprt
capi.init(); st = capi.stream(); P = ctypes.c_void_p
pl = capi.probe_lib()
n = 1 << 26
rng = np.random.default_rng(0)
o0, o1, s = capi.Buf(np.float32, n), capi.Buf(np.float32, n), capi.Buf(np.float32, n)
x = capi.Buf.from_numpy(rng.uniform(-1, 1, n).astype(np.float32))
names = {0: "2 x 4-B lookups, a and b written", 1: "1 x 8-B lookup, a and b written", 2: "2 x 4-B lookups -> fma",
         3: "1 x 8-B lookup -> fma", 4: "a streamed, b looked up -> fma", 5: "a looked up, c streamed -> fma"}
print(f"# tools/probe_gather_pair.py on 1 x MI355X: {n >> 20} Mi elements, random indices into K-entry tables")
for logk in (19, 20, 21, 22):
    K = 1 << logk
    A = capi.Buf.from_numpy(rng.uniform(-1, 1, K).astype(np.float32))
    B = capi.Buf.from_numpy(rng.uniform(-1, 1, K).astype(np.float32))
    AB = capi.Buf(np.float32, 2 * K)
    idx = capi.Buf.from_numpy(rng.integers(0, K, n).astype(np.uint32))
    capi.check(pl.ek_hip_probe_interleave(P(AB.ptr), P(A.ptr), P(B.ptr), ctypes.c_size_t(K)))
    fns = {v: (lambda v=v: capi.check(pl.ek_hip_probe_gather_pair(v, P(o0.ptr), P(o1.ptr), P(A.ptr), P(B.ptr), P(AB.ptr), P(x.ptr),
                                                                   P(s.ptr), P(idx.ptr), ctypes.c_size_t(n)))) for v in range(6)}
    for S in (2, 4):
        fns[f"1 x 8-B lookup -> fma, {S} table slices in time"] = lambda S=S: capi.check(pl.ek_hip_probe_gather_pair_sliced(
            S, P(o0.ptr), P(AB.ptr), ctypes.c_size_t(K), P(x.ptr), P(idx.ptr), ctypes.c_size_t(n)))
    fns["interleave"] = lambda: capi.check(pl.ek_hip_probe_interleave(P(AB.ptr), P(A.ptr), P(B.ptr), ctypes.c_size_t(K)))

    def prod():
        capi.ternary("fmadd", capi.gather(A, idx), x, capi.gather(B, idx))
    fns["product: gather, gather, fmadd"] = prod
    samples = {k: [] for k in fns}
    for _ in range(5):
        for k, f in fns.items():
            samples[k].append(hiprt.time_region(st, f, iters=10, warmup=2))
    # correctness of the fused variants against numpy on a sample
    hA, hB, hx, hi = A.numpy(), B.numpy(), x.numpy()[:4096], idx.numpy()[:4096]
    fns[3]()
    capi.sync()
    ref = (hA[hi].astype(np.float64) * hx + hB[hi]).astype(np.float32)
    assert np.allclose(o0.numpy()[:4096], ref, rtol=1e-6, atol=1e-6)
    for k, v in samples.items():
        ms = statistics.median(v)
        label = names.get(k, k)
        print(f"K=2^{logk} ({K * 4 >> 10:6d} KiB per table)  {label:52s} {ms:7.4f} ms")
