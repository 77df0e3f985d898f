"""cfg3a as a PIPELINE of probe kernels (producer -> consumer chains like the real tape) under different cache
policies, because a single kernel re-reading the same buffers is flattered by Infinity-Cache hits.
stage = (body, inputs, outputs); policy = per-stage (U, ntl, nts)."""
import ctypes, itertools, os, statistics, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enoki_amd import capi, hiprt
capi.init(); st = capi.stream()
n = int(os.environ.get("PROBE_N", 1 << 26))
rng = np.random.default_rng(0)
h = rng.uniform(-1, 1, n).astype(np.float32)
B = {k: capi.Buf.from_numpy(h) if k in "axb" else capi.Buf(np.float32, n) for k in ["a", "x", "b", "u", "s", "c", "gu", "ga", "gb"]}
P = ctypes.c_void_p
def launch(body, u, ntl, nts, o0, o1, i0, i1=None, i2=None):
    capi.check(capi.probe_lib().ek_hip_probe(body, u, ntl, nts, 0, P(B[o0].ptr), P(B[o1].ptr) if o1 else None, P(B[i0].ptr),
                                     P(B[i1].ptr) if i1 else None, P(B[i2].ptr) if i2 else None, ctypes.c_size_t(n)))
# stages of cfg3a: fmadd -> sincos -> hsum(read) -> scale -> mul2 -> scale ; 60 B/elt
def step(pol):
    launch(1, *pol["multi3"], "u", None, "a", "x", "b")
    launch(2, *pol["sincos"], "s", "c", "u")
    launch(3, *pol["read"], "gb", None, "s")
    launch(4, *pol["single"], "gu", None, "c")
    launch(5, *pol["multi2"], "ga", None, "x", "gu")
    launch(4, *pol["single"], "gb", None, "gu")
cands = {
  "all-nt U1 (previous production)": dict(multi3=(1,1,1), sincos=(1,1,1), read=(1,1,1), single=(1,1,1), multi2=(1,1,1)),
  "probe-optimal (current)":         dict(multi3=(1,1,0), sincos=(2,0,1), read=(1,1,1), single=(1,0,1), multi2=(1,1,0)),
  "ntl everywhere, nts only single": dict(multi3=(1,1,0), sincos=(1,1,1), read=(1,1,1), single=(1,1,1), multi2=(1,1,0)),
  "ntl everywhere, no nts":          dict(multi3=(1,1,0), sincos=(1,1,0), read=(1,1,0), single=(1,1,0), multi2=(1,1,0)),
  "all-nt, sincos U2":               dict(multi3=(1,1,1), sincos=(2,1,1), read=(1,1,1), single=(1,1,1), multi2=(1,1,1)),
  "all-nt, U2 single+sincos":        dict(multi3=(1,1,1), sincos=(2,1,1), read=(1,1,1), single=(2,1,1), multi2=(1,1,1)),
  "no nt at all":                    dict(multi3=(1,0,0), sincos=(1,0,0), read=(1,0,0), single=(1,0,0), multi2=(1,0,0)),
  "plain loads, nt stores":          dict(multi3=(1,0,1), sincos=(1,0,1), read=(1,0,1), single=(1,0,1), multi2=(1,0,1)),
  "all-nt U2 everywhere":            dict(multi3=(2,1,1), sincos=(2,1,1), read=(2,1,1), single=(2,1,1), multi2=(2,1,1)),
}
samples = {k: [] for k in cands}
for rep in range(5):
    for k, pol in cands.items():
        samples[k].append(hiprt.time_region(st, lambda: step(pol), iters=10, warmup=2))
print(f"# cfg3a pipeline emulation, n = {n}, 60 B/elt; median ms per step, TB/s, % of 8 TB/s")
for k, v in sorted(samples.items(), key=lambda kv: statistics.median(kv[1])):
    ms = statistics.median(v)
    print(f"{k:36s} {ms:7.4f} ms  {60 * n / ms / 1e9:6.3f} TB/s  {60 * n / ms / 1e9 / 8 * 100:5.1f}%")
