import sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from enoki_amd import capi, synth
import enoki_amd.hip as ek
capi.init()
n = 1 << 26
vals = synth.uniform_pm1(0, n, 3)
for logk in (12, 13, 14, 15, 16):
    K = 1 << logk
    idx = synth.index_mod(0, n, 4, K); t = ek.Float32.zero(K)
    ek.scatter_add(t, vals, idx)
    capi.profile_begin()
    for _ in range(3): ek.scatter_add(t, vals, idx)
    for k in capi.profile_end(): print(logk, k["kernel"], k["launches"], round(k["total_ms"] / k["launches"], 4))
