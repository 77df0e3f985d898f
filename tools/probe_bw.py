"""Sweep the streaming-kernel design space on the GPU (csrc/probe.hip) and time the production
kernels per launch.  Usage (GPU box):  python tools/probe_bw.py [--n 67108864] > gpurun_out/probe.txt

Every configuration is timed `--reps` times in round-robin order (so clock / cache state drift hits
all configurations alike) and the MEDIAN is reported."""
import argparse
import ctypes
import itertools
import os
import statistics
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enoki_amd import capi, hiprt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1 << 26)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--sweep", action="store_true", help="run the (U, nt, grid) design-space sweep")
args = ap.parse_args()
n = args.n
capi.init()
st = capi.stream()
rng = np.random.default_rng(0)
host = rng.uniform(-1, 1, n).astype(np.float32)
a = capi.Buf.from_numpy(host); b = capi.Buf.from_numpy(host[::-1].copy()); c = capi.Buf.from_numpy(host)
o0 = capi.Buf(np.float32, n); o1 = capi.Buf(np.float32, n)
bytes_per_elt = {0: 8, 1: 16, 2: 12, 3: 4, 4: 8, 5: 12, 6: 8}
names = {0: "copy", 1: "fmadd", 2: "sincos", 3: "read", 4: "scale", 5: "mul2", 6: "sin"}
P = ctypes.c_void_p


def median_ms(fns, iters, reps):
    """fns: dict key -> callable; returns dict key -> median ms per call"""
    samples = {k: [] for k in fns}
    for _ in range(reps):
        for k, f in fns.items():
            samples[k].append(hiprt.time_region(st, f, iters=iters, warmup=2))
    return {k: statistics.median(v) for k, v in samples.items()}


print(f"# n = {n} f32 elements ({n * 4 / 2**20:.0f} MiB per array); TB/s of ALGORITHMIC bytes; HBM peak 8.0 TB/s")

if args.sweep:
    fns = {}
    for body in [0, 1, 2, 3, 4, 5, 6]:
        for u, (ntl, nts), bpc in itertools.product([1, 2, 4], [(0, 0), (0, 1), (1, 0), (1, 1)], [0]):
            def f(body=body, u=u, ntl=ntl, nts=nts, bpc=bpc):
                capi.check(capi.probe_lib().ek_hip_probe(body, u, ntl, nts, bpc, P(o0.ptr), P(o1.ptr), P(a.ptr), P(b.ptr),
                                                 P(c.ptr), ctypes.c_size_t(n)))
            fns[(body, u, ntl, nts, bpc)] = f
    res = median_ms(fns, args.iters, args.reps)
    print("body     U ntl nts bpc    ms     TB/s")
    best = {}
    for (body, u, ntl, nts, bpc), ms in res.items():
        tbs = bytes_per_elt[body] * n / ms / 1e9
        print(f"{names[body]:7s} {u:2d} {ntl:3d} {nts:3d} {bpc:3d} {ms:7.4f} {tbs:7.3f}")
        if tbs > best.get(body, (0,))[0]:
            best[body] = (tbs, u, ntl, nts, bpc)
    print("# best per body (bpc 0 = one-shot grid)")
    for body, (tbs, u, ntl, nts, bpc) in best.items():
        print(f"# {names[body]:7s} {tbs:6.3f} TB/s  ({tbs / 8 * 100:.1f}% of 8 TB/s)  U={u} ntl={ntl} nts={nts} bpc={bpc}")

K = 1 << 20
idx_host = (rng.integers(0, K, n)).astype(np.uint32)
idx = capi.Buf.from_numpy(idx_host)
table = capi.Buf.from_numpy(rng.uniform(-1, 1, K).astype(np.float32))
table8 = capi.fill(np.float32, 0.0, 8 * K)
keep = []
prod = {
    ("fmadd(a,x,b)", 16): lambda: keep.append(capi.ternary("fmadd", a, b, c)) or keep.clear(),
    ("sincos(a)", 12): lambda: keep.append(capi.sincos(a)) or keep.clear(),
    ("sin(a)", 8): lambda: keep.append(capi.unary("sin", a)) or keep.clear(),
    ("exp(a)", 8): lambda: keep.append(capi.unary("exp", a)) or keep.clear(),
    ("safe_mul(c, imm)", 8): lambda: keep.append(capi.binary("safe_mul", a, 1.0)) or keep.clear(),
    ("safe_mul(x, g)", 12): lambda: keep.append(capi.binary("safe_mul", a, b)) or keep.clear(),
    ("safe_fmadd(w,g,acc)", 16): lambda: keep.append(capi.ternary("safe_fmadd", a, b, c)) or keep.clear(),
    ("gather(K=1Mi)", 12): lambda: keep.append(capi.gather(table, idx)) or keep.clear(),
    ("scatter_add(K=1Mi)", 8): lambda: capi.scatter_add(table, a, idx),
    ("hsum(a)", 4): lambda: keep.append(capi.reduce("hsum", a)) or keep.clear(),
    ("hsum_safe_mul(w,g)", 8): lambda: keep.append(capi.hsum_safe_mul(a, b)) or keep.clear(),
    ("probe scatter_add shared", 8): lambda: capi.check(capi.probe_lib().ek_hip_probe_scatter_add(
        0, P(table8.ptr), ctypes.c_size_t(K), None, P(a.ptr), P(idx.ptr), ctypes.c_size_t(n))),
    ("probe scatter_add per-XCD+fold", 8): lambda: capi.check(capi.probe_lib().ek_hip_probe_scatter_add(
        1, P(table8.ptr), ctypes.c_size_t(K), P(table.ptr), P(a.ptr), P(idx.ptr), ctypes.c_size_t(n))),
}
res = median_ms(prod, args.iters, args.reps)
print("# production kernels: median ms per launch")
for (name, bpe), ms in res.items():
    tbs = bpe * n / ms / 1e9
    print(f"prod {name:32s} {ms:7.4f} ms  {tbs:6.3f} TB/s  ({tbs / 8 * 100:5.1f}%)  {n / ms / 1e6:7.1f} Gelem/s")

# correctness of the per-XCD scatter_add experiment (exact in integers-as-floats)
ones = capi.fill(np.float32, 1.0, n)
t8 = capi.fill(np.float32, 0.0, 8 * K); tout = capi.fill(np.float32, 0.0, K)
capi.check(capi.probe_lib().ek_hip_probe_scatter_add(1, P(t8.ptr), ctypes.c_size_t(K), P(tout.ptr), P(ones.ptr), P(idx.ptr),
                                             ctypes.c_size_t(n)))
ok = np.array_equal(tout.numpy(), np.bincount(idx_host, minlength=K).astype(np.float32))
per_copy = t8.numpy().reshape(8, K).sum(axis=1)
print(f"# per-XCD scatter_add histogram exact: {ok}; elements landed per XCD copy: {per_copy.astype(np.int64).tolist()}")
