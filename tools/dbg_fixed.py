import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from enoki_amd import capi
from conftest import uniform_pm1
capi.init()
def up(a): return capi.Buf.from_numpy(a)
K, n = 1 << 18, (1 << 20) + 11
rng = np.random.default_rng(12)
A = uniform_pm1(K, 13).astype(np.float32); C = uniform_pm1(K, 14).astype(np.float32)
for xscale in (1e-20, 64.0):
    x = (uniform_pm1(n, 15) * xscale).astype(np.float32); idx = rng.integers(0, K, n).astype(np.uint32)
    dA, dC, dx, di = up(A), up(C), up(x), up(idx)
    u = capi.map_gathered("fmadd", capi.G(dA, di), dx, capi.G(dC, di))
    kept32 = capi.unary("sin", u).numpy(); w32 = (kept32 * x).astype(np.float32)
    for hints in (3, 1):
        b = capi.Bucketed("fmadd", dA, dx, dC, di, hints=hints)
        y = b.reduce("hsum", "cos", keep=True, keep_op="sin").numpy()[0]
        g0, g1 = up(np.zeros(K, np.float32)), up(np.zeros(K, np.float32))
        b.scatter_add([g0, g1], [("sin", 0, False), ("sin", 0, True)], fresh=[1, 1])
        for name, g, t in (("g0", g0.numpy(), kept32.astype(np.float64)), ("g1", g1.numpy(), w32.astype(np.float64))):
            ref = np.bincount(idx, weights=t, minlength=K); sabs = np.bincount(idx, weights=np.abs(t), minlength=K); cnt = np.bincount(idx, minlength=K)
            err = np.abs(g - ref); bound = 2.0**-24 * (cnt + 1) * sabs + 1e-300
            k = int(np.argmax(err / bound))
            print(xscale, "hints", hints, name, "worst ratio", (err / bound)[k], "entry", k, "cnt", cnt[k], "g", g[k], "ref", ref[k], "sabs", sabs[k], "terms", t[idx == k][:6])
        b.destroy()
