"""hsum(f(fmadd(a, x, b))) forward + backward() on leaf arrays for the second-wave functions (round 6: their derivative weights are one
map of the argument, so the step stays two chain kernels); Gelem/s per function.  python tools/probe_second_wave.py [log2 n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import enoki_amd.hip_autodiff as ad
from enoki_amd import synth
ad.hip_init(0)
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 26)
a0, x0, b0 = synth.uniform_pm1(0, n, 1), synth.uniform_pm1(0, n, 2), synth.uniform_pm1(0, n, 3)
xd = ad.Float32(x0)
for name in ("sin", "tanh", "tan", "atan", "sinh", "cosh"):
    f = getattr(ad, name)
    def step():
        a, b = ad.Float32(a0), ad.Float32(b0)
        ad.set_requires_gradient(a); ad.set_requires_gradient(b)
        y = ad.hsum(f(ad.fmadd(a, xd, b)))
        ad.backward(y)
        return ad.gradient(a), ad.gradient(b)
    for _ in range(5): step()
    ad.hip_sync()
    k0 = ad.hip_launch_count()
    reps, dt = 10, 1e9
    for _ in range(3):                       # (best of three batches: the first batch of a process can meet the allocator growing its pools)
        t = time.perf_counter()
        for _ in range(reps): step()
        ad.hip_sync()
        dt = min(dt, (time.perf_counter() - t) / reps)
    reps = 3 * reps
    import json
    ad.hip_profile_begin()
    for _ in range(5): step()
    prof = json.loads(ad.hip_profile_end())
    ks = "  ".join(f"{k['kernel']} {k['total_ms'] / k['launches'] * 1e3:.0f} us" for k in prof if k["launches"] and k["total_ms"] / k["launches"] > 0.02)
    print(f"{name:5s} {n / dt / 1e9:7.1f} Gelem/s  {dt * 1e3:.4f} ms/step  {(ad.hip_launch_count() - k0) / reps:.1f} launches per step  ({32 * n / dt / 1e12:.2f} TB/s on 32 B/elt)   {ks}")
