"""Per-launch time and HBM fraction of the transcendental kernels (f32 second wave, f64).  GPU box:
python tools/probe_math.py > gpurun_out/probe_math.txt"""
import os
import statistics
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enoki_amd import capi, hiprt  # noqa: E402

capi.init()
st = capi.stream()
n = 1 << 25
rng = np.random.default_rng(0)
keep = []
print(f"# n = {n} elements; algorithmic bytes = input + output; HBM peak 8.0 TB/s")
for dt, ops in ((np.float32, ["sin", "exp", "log", "tan", "asin", "acos", "atan", "sinh", "tanh", "asinh", "acosh", "atanh", "cbrt"]),
                (np.float64, ["sin", "cos", "exp", "log", "tan", "asin", "atan", "sinh", "tanh", "asinh", "atanh", "cbrt"])):
    a = capi.Buf.from_numpy(rng.uniform(0.1, 0.9, n).astype(dt))
    b = capi.Buf.from_numpy(rng.uniform(1.1, 3.0, n).astype(dt))
    for op in ops:
        src = b if op == "acosh" else a
        f = lambda op=op, src=src: keep.append(capi.unary(op, src)) or keep.clear()
        ms = statistics.median(hiprt.time_region(st, f, iters=10, warmup=2) for _ in range(3))
        bytes_ = 2 * n * np.dtype(dt).itemsize
        print(f"{np.dtype(dt).name:8s} {op:6s} {ms:7.4f} ms  {bytes_ / ms / 1e9:6.3f} TB/s  ({bytes_ / ms / 1e9 / 8 * 100:5.1f}% of HBM peak)")
    for op in ["atan2", "pow"]:
        f = lambda op=op: keep.append(capi.binary(op, a, b)) or keep.clear()
        ms = statistics.median(hiprt.time_region(st, f, iters=10, warmup=2) for _ in range(3))
        bytes_ = 3 * n * np.dtype(dt).itemsize
        print(f"{np.dtype(dt).name:8s} {op:6s} {ms:7.4f} ms  {bytes_ / ms / 1e9:6.3f} TB/s  ({bytes_ / ms / 1e9 / 8 * 100:5.1f}% of HBM peak)")
