"""Why does fill() take 83 us inside the cfg4 step when the stand-alone probe says 24 us?  (tools/, run on the GPU box)"""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import enoki_amd.hip as ek
ek.hip_init(0)
n = 1 << 25
F = ek.Float32


def prof(label, fn, reps=10):
    fn(); ek.hip_sync()
    ek.hip_profile_begin()
    for _ in range(reps):
        fn()
    p = json.loads(ek.hip_profile_end())
    for k in p:
        if k["launches"]:
            print(f"{label:40s} {k['kernel']:20s} x{k['launches']:3d}  {k['total_ms'] / k['launches'] * 1e3:8.1f} us")


keep = {}
def a():
    keep["x"] = F.full(-1.0, n)
prof("full, ping-pong blocks", a)
def b():
    keep["x"] = None
    keep["x"] = F.full(-1.0, n)
prof("full, same block", b)
def c():
    keep["x"] = F.full(-1.0, n)
    ek.hip_sync()
prof("full + host sync", c)
def d():
    keep["x"] = F.full(-1.0, n)
    keep["y"] = keep["x"] + F(1.0)
    ek.hip_sync()
prof("full + add + host sync", d)
def e():
    keep["x"] = F.zero(n) + F(0.0)
prof("zero", e)
