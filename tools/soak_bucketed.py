"""Soak test of the bucket-ordered path with the early adjoint (ek_hip_bucketed_pair_create_hinted): random table sizes, lookup
counts, index distributions and element types; hinted and unhinted objects on the same inputs.
  exact part   A = C = 0 (u = 0, cos u = 1): the gradient tables are exact counts / exact sums of an integer-valued x -- any lost,
               doubled or misplaced update shows;
  bounded part random tables: y and both gradient tables against float64 with the class-D bounds of tests/test_bucketed_gpu.py.
GPU box: python tools/soak_bucketed.py [rounds] > gpurun_out/soak_bucketed.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enoki_amd import capi  # noqa: E402

capi.init()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(12345)
EPS = {np.float32: 2.0 ** -24, np.float64: 2.0 ** -53}
bad = 0
up = capi.Buf.from_numpy


def indices(kind, K, n):
    if kind == "uniform":
        return rng.integers(0, K, n)
    if kind == "zipf":
        return np.minimum(rng.zipf(1.3, n) - 1, K - 1)
    if kind == "one bucket":
        return rng.integers(0, min(K, 3000), n) + (K // 2 if K > 6000 else 0)
    if kind == "few bins":
        return rng.choice(rng.integers(0, K, 7), n)
    return np.sort(rng.integers(0, K, n))                      # "sorted": long runs per bucket


for r in range(rounds):
    dtype = np.float32 if r % 3 else np.float64
    bins = 16384 if dtype == np.float32 else 8192
    K = int(rng.integers(bins + 1, 250 * bins if r % 7 == 0 else 100 * bins))
    n = int(rng.integers(1 << 18, 1 << 22))
    kind = ("uniform", "zipf", "one bucket", "few bins", "sorted")[r % 5]
    idx = indices(kind, K, n).astype(np.uint32)
    ii = idx.astype(np.int64)
    cnt = np.bincount(ii, minlength=K)
    # (round 5: the product-then-sum forms of `a * x + b` written with operators ride along)
    op = ("fmadd", "fmsub", "fnmadd", "fnmsub", "muladd", "mulsub", "nmuladd")[r % 7]
    half, other = (("sin", "cos"), ("cos", "sin"))[(r // 2) % 2]
    di = up(idx)
    # ---- exact part
    xi = rng.integers(-2, 3, n).astype(dtype)
    only = os.environ.get("SOAK_ONLY")          # SOAK_ONLY=39[,repeats]: the data of every round are drawn, only that round runs
    if only and r != int(only.split(",")[0]):
        rng.uniform(-1, 1, K); rng.uniform(-1, 1, K); rng.uniform(-1, 1, n)
        continue
    repeats = int(only.split(",")[1]) if only and "," in only else 1
    dz, dxi = up(np.zeros(K, dtype)), up(xi)
    for hints in (capi.Bucketed.HINT_ADJOINT, 0):
        b = capi.Bucketed(op, dz, dxi, dz, di, hints=hints)
        b.reduce("hsum", "sin", keep=True, keep_op="cos")
        g1, gx = capi.fill(dtype, 0, K), capi.fill(dtype, 0, K)
        b.scatter_add([g1, gx], [("cos", 0, False), ("cos", 0, True)])
        ok = (np.array_equal(g1.numpy().astype(np.float64), cnt.astype(np.float64)) and
              np.array_equal(gx.numpy().astype(np.float64), np.bincount(ii, weights=xi.astype(np.float64), minlength=K)))
        b.destroy()
        if not ok:
            bad += 1
            print(f"MISMATCH exact part: round {r} {dtype.__name__} K={K} n={n} {kind} {op} hints={hints}", flush=True)
    # ---- bounded part
    A = rng.uniform(-1, 1, K).astype(dtype); C = rng.uniform(-1, 1, K).astype(dtype); x = rng.uniform(-1, 1, n).astype(dtype)
    dA, dC, dx = up(A), up(C), up(x)
    u = capi.map_gathered(op, capi.G(dA, di), dx, capi.G(dC, di))
    red = capi.unary(half, u).numpy().astype(np.float64)
    kept = capi.unary(other, u).numpy().astype(np.float64)
    for hints in (capi.Bucketed.HINT_ADJOINT, 0) * repeats:
        b = capi.Bucketed(op, dA, dx, dC, di, hints=hints)
        y = float(b.reduce("hsum", half, keep=True, keep_op=other).numpy()[0])
        gk, gxk = capi.fill(dtype, 0, K), capi.fill(dtype, 0, K)
        b.scatter_add([gxk, gk], [(other, 0, True), (other, 0, False)])
        b.destroy()
        depth = max(32768, n // 256 + 1) // 4096 + 24
        ok = abs(y - red.sum()) <= EPS[dtype] * (depth * np.abs(red).sum() + 4 * n)
        why = "" if ok else f" y off by {abs(y - red.sum()):.3e} (bound {EPS[dtype] * (depth * np.abs(red).sum() + 4 * n):.3e})"
        for name, got, terms in (("g", gk, kept), ("gx", gxk, kept * x.astype(np.float64))):
            tr = terms.astype(dtype).astype(np.float64)
            # (float64: the yardstick is itself a float64 sum in ANOTHER order, so two summation errors can add up -- round 39
            # of the round-4 run: three terms, device = one of their three possible sums, numpy = another, 2 ulp apart)
            bound = EPS[dtype] * ((2 if dtype == np.float64 else 1) * cnt * np.bincount(ii, weights=np.abs(tr), minlength=K)) + 1e-300
            err = np.abs(got.numpy().astype(np.float64) - np.bincount(ii, weights=tr, minlength=K))
            if not bool((err <= bound).all()):
                ok = False
                w = int(np.argmax(err / bound))
                why += f" {name}: {int((err > bound).sum())} entries, worst entry {w} (count {cnt[w]}) err {err[w]:.3e} bound {bound[w]:.3e}"
                if cnt[w] <= 4 and os.environ.get("SOAK_ONLY"):
                    import itertools
                    tw = tr[ii == w]
                    sums = sorted({float(np.add.reduce(np.array(p, dtype))) for p in itertools.permutations(tw.astype(dtype))})
                    why += f"\n    terms {[float(t).hex() for t in tw]} device {float(got.numpy()[w]).hex()} orders {[v.hex() for v in sums]}"
                    why += f"\n    u {[float(v).hex() for v in u.numpy()[ii == w]]} x {[float(v).hex() for v in x[ii == w]]}"
        if not ok:
            bad += 1
            print(f"MISMATCH bounded part: round {r} {dtype.__name__} K={K} n={n} {kind} {op} {half} hints={hints}:{why}", flush=True)
    print(f"round {r:3d} {dtype.__name__:8s} K={K:8d} n={n:8d} {kind:10s} {op:7s} hsum({half}) ok", flush=True)
print("soak result:", "FAILED" if bad else f"{rounds} rounds, hinted and unhinted: exact part exact, bounded part inside its bounds")
