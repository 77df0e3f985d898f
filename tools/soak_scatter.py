"""Soak test of the LDS exchange-lock accumulate: many repetitions with integer-valued floats (exact sums), uniform
and skewed indices; any lost or duplicated update shows up as a count mismatch.  GPU box: python tools/soak_scatter.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enoki_amd import capi  # noqa: E402

capi.init()
rng = np.random.default_rng(0)
n, reps = 1 << 22, int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0
for K in (1 << 12, 1 << 14, 1 << 20, 1 << 23):
    for name, idx in (("uniform", rng.integers(0, K, n).astype(np.uint32)),
                      ("zipf", np.minimum(rng.zipf(1.2, n) - 1, K - 1).astype(np.uint32))):
        want = np.bincount(idx, minlength=K).astype(np.int64)
        v = capi.Buf.from_numpy(np.ones(n, np.float32)); i = capi.Buf.from_numpy(idx)
        for r in range(reps):
            t = capi.fill(np.float32, 0.0, K)
            capi.scatter_add(t, v, i)
            if r % 10 == 9 or r == reps - 1:
                got = t.numpy().astype(np.int64)
                if not np.array_equal(got, want):
                    bad += 1
                    print(f"MISMATCH K={K} {name} rep {r}: {np.abs(got - want).sum()} updates off", flush=True)
        print(f"K=2^{K.bit_length() - 1:2d} {name:8s} {reps} repetitions ok", flush=True)
# 64-bit elements (8 Ki-bin buckets, 64-bit exchange lock) and the multi-table path with a fused weight
for K in (1 << 13, 1 << 20):
    idx = rng.integers(0, K, n).astype(np.uint32)
    want = np.bincount(idx, minlength=K).astype(np.int64)
    i = capi.Buf.from_numpy(idx)
    v64 = capi.Buf.from_numpy(np.ones(n, np.float64)); v32 = capi.Buf.from_numpy(np.ones(n, np.float32))
    w32 = capi.Buf.from_numpy(np.full(n, 2.0, np.float32))
    for r in range(reps):
        t64 = capi.fill(np.float64, 0.0, K)
        capi.scatter_add(t64, v64, i)
        ta, tb = capi.fill(np.float32, 0.0, K), capi.fill(np.float32, 0.0, K)
        capi.scatter_add_multi([ta, tb], [v32, v32], i, weights=[None, w32])
        if r % 10 == 9 or r == reps - 1:
            ok = (np.array_equal(t64.numpy().astype(np.int64), want) and np.array_equal(ta.numpy().astype(np.int64), want)
                  and np.array_equal(tb.numpy().astype(np.int64), 2 * want))
            if not ok:
                bad += 1
                print(f"MISMATCH (f64 / multi) K={K} rep {r}", flush=True)
    print(f"K=2^{K.bit_length() - 1:2d} float64 + two-table weighted {reps} repetitions ok", flush=True)
print("soak result:", "FAILED" if bad else "all exact")
