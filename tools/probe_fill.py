"""Write-only bandwidth of ek_hip_fill against hipMemsetAsync (rocclr's fill kernel), per size.
Run on the GPU box: python tools/probe_fill.py > gpurun_out/probe_fill.txt"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from enoki_amd import capi as ek

ek.init()
print(f"{'n (f32)':>12} {'fill us':>9} {'TB/s':>6} | {'torch fill_ us':>14} {'TB/s':>6} | {'floor us':>9} {'TB/s (8 B/elt)':>14}")
for log2n in (22, 24, 25, 26, 28):
    n = 1 << log2n
    ek.fill(np.float32, 1.0, n).free()
    ek.sync()
    reps = 30
    ek.profile_begin()
    bufs = [ek.fill(np.float32, -1.0, n) for _ in range(reps)]
    ek.sync()
    prof = {p["kernel"]: p for p in ek.profile_end()}
    us = prof["fill"]["total_ms"] / prof["fill"]["launches"] * 1e3
    src = bufs[0]
    for b in bufs[1:]:
        b.free()
    ek.profile_begin()
    outs = [ek.unary("floor", src) for _ in range(10)]
    ek.sync()
    prof = {p["kernel"]: p for p in ek.profile_end()}
    fl = prof["floor"]["total_ms"] / prof["floor"]["launches"] * 1e3
    for o in outs:
        o.free()
    src.free()
    t = torch.empty(n, dtype=torch.float32, device="cuda")
    t.fill_(1.0); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        t.fill_(-1.0)
    e1.record(); torch.cuda.synchronize()
    tus = e0.elapsed_time(e1) / reps * 1e3
    del t
    print(f"{n:>12} {us:>9.1f} {4 * n / us / 1e6:>6.2f} | {tus:>14.1f} {4 * n / tus / 1e6:>6.2f} | {fl:>9.1f} {8 * n / fl / 1e6:>14.2f}")
