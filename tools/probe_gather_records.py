"""Struct gather: staged {x, y, ..} records (one request per element) against the plain kernels (one request per component),
per table size and lookup count.  Times are per call INCLUDING the staging pass.
Run on the GPU box: python tools/probe_gather_records.py > gpurun_out/probe_gather_records.txt"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from enoki_amd import capi as ek

ek.init()
rng = np.random.default_rng(1)


def timed(tables, di, reps=6):
    for o in ek.gather_multi(tables, di):
        o.free()
    ek.sync()
    ek.profile_begin()
    for _ in range(reps):
        for o in ek.gather_multi(tables, di):
            o.free()
    ek.sync()
    prof = ek.profile_end()
    return sum(p["total_ms"] for p in prof) / reps * 1e3, "+".join(sorted({p["kernel"] for p in prof}))


print(f"{'C':>2} {'K':>10} {'n':>10} | {'plain us':>9} {'kernels':<28} | {'records us':>10} {'stage us':>8} | speedup | default plan")
ONE = "--one" in sys.argv          # a single shape, for counter runs: tools/profile_gather_records.sh
for count in ((3,) if ONE else (2, 3, 4)):
    for log2k in ((23,) if ONE else (18, 20, 21, 22, 23, 25)):
        k = 1 << log2k
        tables = [ek.Buf.from_numpy(rng.standard_normal(k).astype(np.float32)) for _ in range(count)]
        for log2n in ((24,) if ONE else (20, 22, 24, 26)):
            n = 1 << log2n
            di = ek.Buf.from_numpy(rng.integers(0, k, n).astype(np.uint32))
            ek.set_tuning("gather_records", 0)
            t_plain, names = timed(tables, di)
            ek.set_tuning("gather_records", 2)
            ek.profile_begin()
            for o in ek.gather_multi(tables, di):
                o.free()
            ek.sync()
            stage = [p["total_ms"] for p in ek.profile_end() if p["kernel"] == "gather_stage_records"][0] * 1e3
            t_rec, _ = timed(tables, di)
            ek.set_tuning("gather_records", 1)
            plan = ek.lib.ek_hip_gather_multi_plan(ek.NP2EK[np.dtype(np.float32)], ek.NP2EK[np.dtype(np.uint32)], count, k, n)
            print(f"{count:>2} {k:>10} {n:>10} | {t_plain:>9.1f} {names:<28} | {t_rec:>10.1f} {stage:>8.1f} | {t_plain / t_rec:>7.2f} | "
                  f"{['per table', 'one launch', 'records'][plan]}{'' if (plan == 2) == (t_rec < t_plain) else '   <-- wrong'}")
            di.free()
        for t in tables:
            t.free()
