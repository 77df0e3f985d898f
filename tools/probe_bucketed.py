"""Bucket-ordered cfg3b step at the C ABI (ek_hip_bucketed_*): per-kernel times of
   create (count / scan / partition) -> reduce(hsum, sin, keep) -> scatter_add(cos(u), x cos(u))
against the element-order pipeline (gather_pair_fmadd -> hsum(sin) -> scatter_add_multi_map) on the same inputs.
GPU box: python tools/probe_bucketed.py [log2 n] [log2 K] > gpurun_out/probe_bucketed.txt"""
import os, sys
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
out = {}


def bucketed(hints=0, key=""):
    b = capi.Bucketed("fmadd", A, x, B, idx, hints=hints)
    out["y"] = b.reduce("hsum", "sin", keep=True, keep_op="cos")      # what DiffArray::sin_ leads to: sincos, cos kept for the adjoint
    gA, gB = capi.fill(np.float32, 0, K), capi.fill(np.float32, 0, K)
    b.scatter_add([gB, gA], [("cos", 0, False), ("cos", 0, True)])
    out["gA" + key], out["gB" + key] = gA, gB
    if key:
        out["y" + key] = out["y"]
    b.destroy()


def early():
    bucketed(capi.Bucketed.HINT_ADJOINT, "_early")


def element_order():
    u = capi.map_gathered("fmadd", capi.G(A, idx), x, capi.G(B, idx))
    out["y_e"] = capi.reduce_map("hsum", "sin", u)
    gA, gB = capi.fill(np.float32, 0, K), capi.fill(np.float32, 0, K)
    capi.scatter_add_multi_map([gB, gA], [u, u], ["cos", "cos"], idx, weights=[None, x])
    out["gA_e"], out["gB_e"] = gA, gB


print(f"# tools/probe_bucketed.py on 1 x MI355X: n = 2^{logn}, K = 2^{logk}, pieces per CU = {os.environ.get('ENOKI_HIP_BUCKET_PIECES_PER_CU', '1')}")
for name, fn in (("early adjoint", early), ("bucket order", bucketed), ("element order", element_order)):
    for _ in range(3):
        fn()
    capi.sync()
    ms = min(hiprt.time_region(st, fn, iters=10, warmup=1) for _ in range(3))
    print(f"{name:14s} {ms:8.4f} ms per step = {n / ms / 1e6:7.2f} Gelem/s")
    capi.profile_begin()
    for _ in range(5):
        fn()
    for k in sorted(capi.profile_end(), key=lambda k: -k["total_ms"]):
        if k["launches"]:
            avg = k["total_ms"] / k["launches"]
            print(f"    {k['kernel']:26s} x{k['launches'] // 5}  {avg:8.4f} ms  {k['bytes'] / k['launches'] / avg / 1e9:6.2f} TB/s")
y, ye = float(out["y"].numpy()[0]), float(out["y_e"].numpy()[0])
dA = np.abs(out["gA"].numpy().astype(np.float64) - out["gA_e"].numpy()).max()
dB = np.abs(out["gB"].numpy().astype(np.float64) - out["gB_e"].numpy()).max()
print(f"y bucket {y!r}  element {ye!r}  |diff| {abs(y - ye):.3e};  max |gA diff| {dA:.3e}  max |gB diff| {dB:.3e}")
yl = float(out["y_early"].numpy()[0])
dA = np.abs(out["gA_early"].numpy().astype(np.float64) - out["gA_e"].numpy()).max()
dB = np.abs(out["gB_early"].numpy().astype(np.float64) - out["gB_e"].numpy()).max()
print(f"y early  {yl!r}  |diff| {abs(yl - ye):.3e};  max |gA diff| {dA:.3e}  max |gB diff| {dB:.3e}")

# the adjoint alone on a kept partition: what the value streams cost
b = capi.Bucketed("fmadd", A, x, B, idx)
b.reduce("hsum", "sin", keep=True)
g = [capi.fill(np.float32, 0, K) for _ in range(2)]
for name, streams in (("cos(u), x cos(u)   [cfg3b]", [("cos", 0, False), ("cos", 0, True)]),
                      ("u, x u             [no map]", [("copy", 0, False), ("copy", 0, True)]),
                      ("1, x               [no u]", [(None, 1.0, False), (None, 1.0, True)]),
                      ("cos(u)             [one table]", [("cos", 0, False)]),
                      ("u                  [one table, no map]", [("copy", 0, False)])):
    fn = lambda: b.scatter_add(g[:len(streams)], streams)
    ms = min(hiprt.time_region(st, fn, iters=10, warmup=2) for _ in range(3))
    print(f"adjoint only: {name:40s} {ms:8.4f} ms (accumulate + fold)")
