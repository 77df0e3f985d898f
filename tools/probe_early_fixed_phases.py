"""Where a workgroup of the fixed-point k_bucket_pair_forward_adjoint spends its time (measurement build: tools/build_variant.py timing
bucketed_early.hip -DEK_EARLY_TIMING, swapped over enoki_amd/libenoki-hip.so).  python tools/probe_early_fixed_phases.py [log2 n] [log2 K]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enoki_amd import capi
capi.init()
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 26
logk = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n, K = 1 << logn, 1 << logk
rng = np.random.default_rng(0)
A = capi.Buf.from_numpy(rng.uniform(-1, 1, K).astype(np.float32))
B = capi.Buf.from_numpy(rng.uniform(-1, 1, K).astype(np.float32))
x = capi.Buf.from_numpy(rng.uniform(-1, 1, n).astype(np.float32))
idx = capi.Buf.from_numpy(rng.integers(0, K, n).astype(np.uint32))


def step():
    b = capi.Bucketed("fmadd", A, x, B, idx, hints=capi.Bucketed.HINT_ADJOINT | capi.Bucketed.HINT_BOUNDED)
    y = b.reduce("hsum", "sin", keep=True, keep_op="cos")
    gA, gB = capi.fill(np.float32, 0, K), capi.fill(np.float32, 0, K)
    b.scatter_add([gB, gA], [("cos", 0, False), ("cos", 0, True)])
    b.destroy()
    return y


lib = capi.lib
if not hasattr(lib, "ek_hip_debug_early_timing"):
    sys.exit("this libenoki-hip.so is not a -DEK_EARLY_TIMING build")
out = (ctypes.c_ulonglong * 32)()
for _ in range(3):
    step()
lib.ek_hip_debug_early_timing(out)
reps = 10
for _ in range(reps):
    step()
lib.ek_hip_debug_early_timing(out)
t = [int(v) for v in out]
wg = max(t[22], 1)
print(f"# k_bucket_pair_forward_adjoint<Fixed>, n = 2^{logn}, K = 2^{logk}: {wg // reps} workgroups per launch; shader cycles of thread 0, mean per workgroup")
for name, k in (("which piece am I", 16), ("slice staged, planes cleared, guards", 17), ("wave 0's walk", 18), ("waiting for the slowest wave", 19),
                ("tables converted and written", 20), ("finish: ticket (+ the reduction in the last)", 21)):
    print(f"  {name:46s} {t[k] / wg:10.1f}")
print(f"  {'ticket atomic alone, mean / slowest':46s} {t[24] / wg:10.1f} / {t[25]}")
print(f"  {'whole workgroup, mean / slowest':46s} {sum(t[16:22]) / wg:10.1f} / {t[23]}")
