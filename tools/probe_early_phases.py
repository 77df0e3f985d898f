"""Where a wave of k_bucket_pair_forward_adjoint spends a step (measurement build: tools/build_variant.py timing bucketed.hip
-DEK_EARLY_TIMING, swapped over enoki_amd/libenoki-hip.so).  python tools/probe_early_phases.py [log2 n] [log2 K]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enoki_amd import capi, hiprt
capi.init(); st = capi.stream()
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 26
logk = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n, K = 1 << logn, 1 << logk
rng = np.random.default_rng(0)
A = capi.Buf.from_numpy(rng.uniform(-1, 1, K).astype(np.float32))
B = capi.Buf.from_numpy(rng.uniform(-1, 1, K).astype(np.float32))
x = capi.Buf.from_numpy(rng.uniform(-1, 1, n).astype(np.float32))
idx = capi.Buf.from_numpy(rng.integers(0, K, n).astype(np.uint32))


def step():
    b = capi.Bucketed("fmadd", A, x, B, idx, hints=capi.Bucketed.HINT_ADJOINT)
    y = b.reduce("hsum", "sin", keep=True, keep_op="cos")
    gA, gB = capi.fill(np.float32, 0, K), capi.fill(np.float32, 0, K)
    b.scatter_add([gB, gA], [("cos", 0, False), ("cos", 0, True)])
    b.destroy()
    return y


lib = capi.lib
if not hasattr(lib, "ek_hip_debug_early_timing"):
    sys.exit("this libenoki-hip.so is not a -DEK_EARLY_TIMING build")
out = (ctypes.c_ulonglong * 16)()
for _ in range(3):
    step()
lib.ek_hip_debug_early_timing(out)
reps = 10
for _ in range(reps):
    step()
lib.ek_hip_debug_early_timing(out)
t = [int(v) for v in out]
waves, steps = t[6], t[4]
names = ["wait for the step's (l16, x) loads", "record reads (4 x ds_read_b64 + wait)", "arithmetic + claim / add / release x 4", "retry round"]
print(f"# k_bucket_pair_forward_adjoint, n = 2^{logn}, K = 2^{logk}: {waves // reps} waves per launch, {steps / waves:.1f} steps per wave (4 elements per lane and step)")
print(f"# cycle counter = s_memtime (100 MHz-independent shader clock); per STEP of one wave, averaged over all waves and {reps} launches")
tot = sum(t[:4])
for k in range(4):
    print(f"  {names[k]:44s} {t[k] / steps:8.1f} cycles   {100.0 * t[k] / tot:5.1f} %")
print(f"  {'sum of the phases':44s} {tot / steps:8.1f} cycles per step;   whole walk of a wave {t[5] / waves:10.1f} cycles = {t[5] / waves / max(steps / waves, 1):.1f} per step")
wg = 256 * reps
print(f"# per WORKGROUP (256 pieces per launch), cycles; reporting waves: 4 of 16")
print(f"  which piece am I (bucket_piece)              {t[10] / waves:10.1f}")
print(f"  table slice staged + tables cleared + barrier {t[11] / waves:9.1f}")
print(f"  main steps                                   {t[7] / waves:10.1f}")
print(f"  complete pages that do not fill a step       {t[8] / waves:10.1f}")
print(f"  partially filled pages                       {t[9] / waves:10.1f}")
print(f"  walk, mean / slowest reporting wave          {t[5] / waves:10.1f} / {t[12]}")
print(f"  end of wave 0's walk -> tables written       {t[13] / wg:10.1f}")
print(f"  whole workgroup, mean / slowest              {t[14] / wg:10.1f} / {t[15]}")
