"""round 5: blocks per CU of the two-stage reductions (ek_hip_set_tuning("reduce_blocks_per_cu", v); the chain reduction and the plain one
share the grid, so deferred and eager evaluation keep the same bits under any setting).  Kernel times from the library's event profile."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import enoki_amd.hip as ek

ek.hip_init(0)
n = 1 << 26
rng = np.random.default_rng(1)
a, x, b = (ek.Float32(rng.uniform(-1, 1, n).astype(np.float32)) for _ in range(3))
u = ek.fmadd(a, x, b); _ = u.numpy()[:1]


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    ek.hip_profile_begin()
    for _ in range(reps):
        fn()
    prof = json.loads(ek.hip_profile_end())
    return {k["kernel"]: round(k["total_ms"] / k["launches"] * 1e3, 1) for k in prof if k["launches"] >= reps and k["total_ms"] / k["launches"] > 0.02}


for rnd in range(2):
    for bpc in (4, 5, 6, 8, 3):
        ek.hip_set_tuning("reduce_blocks_per_cu", bpc)
        r = {}
        r.update({"cfg2 " + k: v for k, v in timed(lambda: ek.hsum(ek.sin(ek.exp(ek.fmadd(a, x, b))))).items()})
        r.update({"sin " + k: v for k, v in timed(lambda: ek.hsum(ek.sin(ek.fmadd(a, x, b)))).items()})
        r.update({"plain " + k: v for k, v in timed(lambda: ek.hsum(u)).items()})
        r.update({"map " + k: v for k, v in timed(lambda: ek.hsum(ek.sin(u))).items()})
        print("blocks/CU", bpc, r, flush=True)
