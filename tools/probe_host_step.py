"""Host side of the cfg3b step: wall time of every python-level call of the step WITHOUT synchronisation (what the CPU spends
before the GPU can start), at a size where the GPU is faster than the host (n = 2^20 by default), plus the step rate that
results.  python tools/probe_host_step.py [log2 n] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import enoki_amd.hip_autodiff as ek
from enoki_amd import synth
ek.hip_init(0)
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
n, K = 1 << logn, 1 << 20
A0, B0 = synth.uniform_pm1(0, K, 6), synth.uniform_pm1(0, K, 7)
x = ek.Float32(synth.uniform_pm1(0, n, 2))
idx = ek.UInt32(synth.index_mod(0, n, 4, K))
names = ["Float32(A0), Float32(B0)", "set_requires_gradient x2", "gather x2", "fmadd", "sin", "hsum", "backward", "detach + gradient x2", "drop"]
acc = [0.0] * len(names)
now = time.perf_counter


def step(timed):
    t = [now()]
    A = ek.Float32(A0); B = ek.Float32(B0); t.append(now())
    ek.set_requires_gradient(A); ek.set_requires_gradient(B); t.append(now())
    a = ek.gather(A, idx); b = ek.gather(B, idx); t.append(now())
    u = ek.fmadd(a, x, b); t.append(now())
    s = ek.sin(u); t.append(now())
    y = ek.hsum(s); t.append(now())
    ek.backward(y); t.append(now())
    r = (ek.detach(y), ek.gradient(A), ek.gradient(B)); t.append(now())
    del A, B, a, b, u, s, y, r; t.append(now())
    if timed:
        for i in range(len(names)):
            acc[i] += t[i + 1] - t[i]


for _ in range(20):
    step(False)
ek.hip_sync() if hasattr(ek, "hip_sync") else None
l0 = ek.hip_launch_count()
t0 = now()
for _ in range(steps):
    step(True)
t1 = now()
launches = (ek.hip_launch_count() - l0) / steps
print(f"# n = 2^{logn}: {(t1 - t0) / steps * 1e6:.1f} us per step issued by the host ({launches:.1f} kernel launches per step)")
for nm, a in zip(names, acc):
    print(f"  {nm:28s} {a / steps * 1e6:7.1f} us")
