"""scatter_add into tables beyond 4 Mi bins (two-level binned path) by index distribution.  GPU box:
python tools/probe_skew_atomics.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enoki_amd import capi, hiprt  # noqa: E402

capi.init(); st = capi.stream()
rng = np.random.default_rng(0)
for n, K in ((1 << 22, 1 << 23), (1 << 26, 1 << 24)):
    vals = np.ones(n, np.float32)
    for name, idx in (("all_zero", np.zeros(n, np.uint32)), ("zipf", np.minimum(rng.zipf(1.3, n) - 1, K - 1).astype(np.uint32)),
                      ("uniform", rng.integers(0, K, n).astype(np.uint32))):
        t = capi.fill(np.float32, 0.0, K); v = capi.Buf.from_numpy(vals); i = capi.Buf.from_numpy(idx)
        ms = hiprt.time_region(st, lambda: capi.scatter_add(t, v, i), iters=2, warmup=1)
        ok = np.array_equal((t.numpy() / 3.0).astype(np.int64), np.bincount(idx, minlength=K)) if n <= (1 << 24) else \
            abs(float(t.numpy().astype(np.float64).sum()) / 3.0 - n) < 1e-3 * n
        print(f"n=2^{n.bit_length() - 1} K=2^{K.bit_length() - 1} {name:10s} {ms:9.3f} ms per call  {n / ms / 1e6:8.2f} G adds/s  ok: {ok}", flush=True)
