"""Stress of the consumers' finish ticket (csrc/ek_bucketed.h: finish_ticket -- a relaxed agent-scope atomic behind a wait for the wave's
stores): a stale per-workgroup partial would be invisible in a benchmark that repeats ONE step (yesterday's partial equals today's), so
here every launch gets NEW values and the reduced value is checked against float64 each time; sizes from all-workgroups-finish-together
to the headline's.  python tools/stress_finish_ticket.py [launches per size]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enoki_amd import capi
capi.init()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
K = 1 << 20
rng = np.random.default_rng(7)
bad = 0
for logn in (20, 22, 23, 24):
    n = 1 << logn
    idx_h = rng.integers(0, K, n).astype(np.uint32)
    idx = capi.Buf.from_numpy(idx_h)
    worst = 0.0
    for r in range(reps):
        a_h = rng.uniform(-1, 1, K).astype(np.float32); b_h = rng.uniform(-1, 1, K).astype(np.float32)
        x_h = (rng.uniform(-1, 1, n) * (1 + r % 3)).astype(np.float32)
        A, B, x = capi.Buf.from_numpy(a_h), capi.Buf.from_numpy(b_h), capi.Buf.from_numpy(x_h)
        for op, keep, hints in (("sin", "cos", capi.Bucketed.HINT_ADJOINT | capi.Bucketed.HINT_BOUNDED), ("exp", "exp", capi.Bucketed.HINT_ADJOINT)):
            bk = capi.Bucketed("fmadd", A, x, B, idx, hints=hints)
            y = bk.reduce("hsum", op, keep=True, keep_op=keep)
            got = float(y.numpy()[0])
            bk.destroy()
            u = a_h[idx_h].astype(np.float64) * x_h + b_h[idx_h]
            want = float(np.sum(np.sin(u) if op == "sin" else np.exp(u)))
            scale = float(np.sum(np.abs(np.sin(u)) if op == "sin" else np.exp(u)))
            err = abs(got - want) / scale
            worst = max(worst, err)
            if err > 2e-6:
                bad += 1
                print(f"MISMATCH n=2^{logn} launch {r} {op}: got {got!r} want {want!r} rel-to-abs-sum {err:.3e}")
    print(f"n = 2^{logn}: {2 * reps} launches, worst |y - y64| / sum|terms| = {worst:.2e}")
print("OK" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
