"""Generate tests/golden/ from the UNMODIFIED reference build (oracle/_ref/libenoki_ref.so).

Run in the dev container (where /root/reference exists):   python tests/golden/make_golden.py
The fixtures are what the GPU box (which has no /root/reference) falls back to when oracle/_ref did not
travel; they also pin oracle/enoki_oracle.c independently of the live comparison in test_oracle_vs_ref.py.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib  # noqa: E402
import tape_lib  # noqa: E402
from conftest import SPECIALS_F32, f32_inputs, f64_inputs, uniform_pm1, hash_u32  # noqa: E402

R = oracle_lib.ref()

# ---- tape programs --------------------------------------------------------------------------------
for name, prog in tape_lib.suite().items():
    v, g = tape_lib.run(tape_lib.ref_fn(), prog)
    out = {"value": v, "n_grads": np.array(len(g))}
    for i, gi in enumerate(g):
        if gi is not None:
            out[f"g{i}"] = gi
    np.savez_compressed(os.path.join(HERE, f"tape_{name}.npz"), **out)

for name, prog in tape_lib.gather_suite().items():
    v, g = tape_lib.run(tape_lib.ref_fn(), prog)
    out = {"value": v, "n_grads": np.array(len(g))}
    for i, gi in enumerate(g):
        if gi is not None:
            out[f"g{i}"] = gi
    np.savez_compressed(os.path.join(HERE, f"tape_gather_{name}.npz"), **out)

# two scatters into one buffer: from the AVX-512 flavour of the reference build (the AVX2 row segfaults in the reference
# itself on this pattern, oracle/Makefile)
if tape_lib.ref512_available():
    for name, prog in tape_lib.scatter_twice_suite().items():
        v, g = tape_lib.run(tape_lib.ref512_fn(), prog)
        out = {"value": v, "n_grads": np.array(len(g))}
        for i, gi in enumerate(g):
            if gi is not None:
                out[f"g{i}"] = gi
        np.savez_compressed(os.path.join(HERE, f"tape_scatter_twice_{name}.npz"), **out)

# ---- class C against the reference's SCALAR row (oracle/Makefile refscalar: rcp = 1 / a, rsqrt = 1 / sqrt(a)) -----------
# rcp / rsqrt / div: what the device must reproduce bit for bit.  tan ... i0e: the second anchor beside the AVX2 row (the two
# rows of the reference differ from each other: unfused vs fused polynomial cores, exact vs approximate reciprocals).
CLASS_C_OPS = ["rcp", "rsqrt", "tan", "cot", "sinh", "cosh", "tanh", "erf", "erfc", "i0e"]
CLASS_C_PROGRAMS = ["div_rcp_rsqrt", "sw_trig", "sw_hyp", "sw_sum", "sw_cbrt_pow"]
if tape_lib.refscalar_available():
    S = oracle_lib.ref_scalar()
    xs = f32_inputs(8192, seed=211, scale=4.0)
    xs = np.concatenate([xs, np.linspace(-12, 12, 4096, dtype=np.float32), uniform_pm1(4096, 5)])
    cc = {"x": xs, "y": np.roll(xs, 7) + np.float32(0.25)}
    for op in CLASS_C_OPS:
        arg = np.abs(xs) + np.float32(1e-3) if op == "rsqrt" else xs
        cc[f"scalar_{op}"] = S.unary(op, arg)
        cc[f"avx2_{op}"] = R.unary(op, arg)
    cc["scalar_div"] = S.binary("div", cc["x"], cc["y"])
    cc["avx2_div"] = R.binary("div", cc["x"], cc["y"])
    np.savez_compressed(os.path.join(HERE, "classc_scalar.npz"), **cc)
    progs = tape_lib.suite()
    for name in CLASS_C_PROGRAMS:
        v, g = tape_lib.run(tape_lib.refscalar_fn(), progs[name])
        out = {"value": v, "n_grads": np.array(len(g))}
        for i, gi in enumerate(g):
            if gi is not None:
                out[f"g{i}"] = gi
        np.savez_compressed(os.path.join(HERE, f"tape_scalar_{name}.npz"), **out)

# ---- elementwise ops on a fixed input set (incl. specials) -------------------------------------------
n = 4096
a = f32_inputs(n, seed=101, scale=20.0); b = f32_inputs(n, seed=102, scale=20.0)[::-1].copy(); c = f32_inputs(n, seed=103)
ops = {"in_a": a, "in_b": b, "in_c": c}
for op in ["neg", "abs", "sqrt", "floor", "ceil", "round", "trunc", "sin", "cos", "exp", "log", "sign"]:
    ops[f"unary_{op}"] = R.unary(op, a)
s, co = R.sincos(a); ops["sincos_s"], ops["sincos_c"] = s, co
for op in ["add", "sub", "mul", "div", "min", "max", "safe_mul"]:
    ops[f"binary_{op}"] = R.binary(op, a, b)
for op in ["fmadd", "fmsub", "fnmadd", "fnmsub", "safe_fmadd"]:
    ops[f"ternary_{op}"] = R.ternary(op, a, b, c)
for op in ["eq", "neq", "lt", "le", "gt", "ge"]:
    ops[f"compare_{op}"] = R.compare(op, a, b)
np.savez_compressed(os.path.join(HERE, "elementwise_f32.npz"), **ops)

# ---- second-wave math (array_math.h tan .. atanh, cbrt, atan2, pow, fmod, ldexp) -----------------------
ops2 = {"in_a": a, "in_b": b, "in_c": c}
for op in ["tan", "cot", "asin", "acos", "atan", "sinh", "cosh", "tanh", "asinh", "acosh", "atanh", "cbrt"]:
    ops2[f"unary_{op}"] = R.unary(op, a)
unit = uniform_pm1(n, seed=104) * np.float32(1.2)          # covers [-1, 1] and a little outside (NaN results)
ops2["in_unit"] = unit
for op in ["asin", "acos", "atanh"]:
    ops2[f"unit_{op}"] = R.unary(op, unit)
for op in ["atan2", "pow", "fmod"]:
    ops2[f"binary_{op}"] = R.binary(op, a, b)
ops2["in_e"] = np.trunc(b).astype(np.float32)
ops2["ldexp"] = R.binary("ldexp", c, np.clip(ops2["in_e"], -100, 100))
np.savez_compressed(os.path.join(HERE, "elementwise2_f32.npz"), **ops2)

# ---- float64 transcendentals (array_math.h double branches) ---------------------------------------------
d = f64_inputs(n, seed=201, scale=20.0, limit=3e9); dpos = np.abs(f64_inputs(n, seed=202, scale=1e3))
ops64 = {"in_d": d, "in_pos": dpos, "sin": R.unary("sin", d), "cos": R.unary("cos", d), "exp": R.unary("exp", d),
         "log": R.unary("log", d), "log_pos": R.unary("log", dpos)}
ops64["sincos_s"], ops64["sincos_c"] = R.sincos(d)
dunit = f64_inputs(n, seed=203, scale=0.6)
ops64["in_unit"] = dunit
for op in ["tan", "cot", "atan", "sinh", "cosh", "tanh", "asinh", "cbrt"]:
    ops64[f"sw_{op}"] = R.unary(op, d)
for op in ["asin", "acos", "atanh"]:
    ops64[f"sw_{op}"] = R.unary(op, dunit)
ops64["sw_acosh"] = R.unary("acosh", dpos)
d2 = f64_inputs(n, seed=204, scale=20.0, limit=3e9)[::-1].copy()
ops64["in_d2"] = d2
for op in ["atan2", "pow", "fmod"]:
    ops64[f"sw_{op}"] = R.binary(op, d, d2)
np.savez_compressed(os.path.join(HERE, "elementwise_f64.npz"), **ops64)

# ---- special functions (include/enoki/special.h:56-312) ------------------------------------------------
sp = {}
srng = np.random.default_rng(77)
for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
    edge = np.array([0, -0.0, 1, -1, 0.5, -0.5, 2, -2, 8, -8, 9, 1e-30, np.inf, -np.inf, np.nan, 3, -3, 4, 5, -5, -2.5], dt)
    wide = np.concatenate([srng.uniform(-12, 12, 4075).astype(dt), edge])
    unit = np.concatenate([srng.uniform(-0.999, 0.999, 4092).astype(dt), np.array([0, -0.0, 1, -1], dt)])
    sp[f"{tag}_wide"], sp[f"{tag}_unit"] = wide, unit
    for op in ["erf", "erfc", "i0e", "dawson", "lgamma", "tgamma"]:
        sp[f"{tag}_{op}"] = R.unary(op, wide)
    for op in ["erfinv", "erfi"]:
        sp[f"{tag}_{op}"] = R.unary(op, unit)
np.savez_compressed(os.path.join(HERE, "special.npz"), **sp)

# ---- Quaternion<FloatX> (include/enoki/quaternion.h), see oracle/ref_driver.cpp:ref_quaternion ------------------
rq = np.random.default_rng(77)
qa = rq.uniform(-1.5, 1.5, (4, 1000)).astype(np.float32); qb = rq.uniform(-1.5, 1.5, (4, 1000)).astype(np.float32)
qt = rq.uniform(0, 1, 1000).astype(np.float32)
qout, qmat = R.quaternion(qa, qb, qt)
np.savez_compressed(os.path.join(HERE, "quaternion.npz"), a=qa, b=qb, t=qt, out=qout, mat=qmat)

# ---- Complex<FloatX> (include/enoki/complex.h), see oracle/ref_driver.cpp:ref_complex ------------------------
ca = uniform_pm1(2 * 2048, 401).reshape(2, 2048) * np.float32(2.5)
cb = uniform_pm1(2 * 2048, 402).reshape(2, 2048) * np.float32(1.5)
np.savez_compressed(os.path.join(HERE, "complex.npz"), a=ca, b=cb, out=R.complex(ca, cb))

# ---- PCG32 (include/enoki/random.h) draw script, see oracle/ref_driver.cpp:ref_pcg32 -------------------
seq = (np.arange(1024, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0xda3e39cb94b95bdb))
pm = ((hash_u32(np.arange(1024, dtype=np.uint64), 5) & np.uint32(3)) != 0).astype(np.uint8)
pc = R.pcg32(0x853c49e6748fea9b, seq, 4, pm, 1000003, -98765)
np.savez_compressed(os.path.join(HERE, "pcg32.npz"), initseq=seq, mask=pm, **pc)

# ---- Matrix<FloatX, N> (include/enoki/matrix.h), see oracle/ref_driver.cpp:ref_matrix ---------------------
mats = {}
for N in (2, 3, 4):
    a = uniform_pm1(N * N * 1024, 300 + N).reshape(N * N, 1024) * np.float32(2)
    b = uniform_pm1(N * N * 1024, 310 + N).reshape(N * N, 1024)
    v = uniform_pm1(N * 1024, 320 + N).reshape(N, 1024)
    for i in range(N):                       # diagonally dominant -> well conditioned inverses
        a[i * N + i] += np.float32(4)
    for k, val in R.matrix(N, a, b, v).items():
        mats[f"m{N}_{k}"] = val
    mats[f"m{N}_a"], mats[f"m{N}_b"], mats[f"m{N}_v"] = a, b, v
np.savez_compressed(os.path.join(HERE, "matrix.npz"), **mats)

# ---- integer ops -----------------------------------------------------------------------------------
rng = np.random.default_rng(7)
iops = {}
for dt in (np.int32, np.uint32):
    info = np.iinfo(dt)
    x = rng.integers(info.min, info.max, n, dtype=dt, endpoint=True); y = rng.integers(info.min, info.max, n, dtype=dt, endpoint=True)
    sh = rng.integers(0, 40, n).astype(dt)
    t = dt.__name__
    iops[f"{t}_x"], iops[f"{t}_y"], iops[f"{t}_sh"] = x, y, sh
    for op in ["neg", "not", "abs", "popcnt", "lzcnt", "tzcnt"]:
        iops[f"{t}_unary_{op}"] = R.unary(op, x)
    for op in ["add", "sub", "mul", "min", "max", "mulhi", "and", "or", "xor"]:
        iops[f"{t}_binary_{op}"] = R.binary(op, x, y)
    for op in ["sl", "sr"]:
        iops[f"{t}_binary_{op}"] = R.binary(op, x, sh)
np.savez_compressed(os.path.join(HERE, "integer.npz"), **iops)

# ---- gather / scatter / reductions / configs ---------------------------------------------------------
K, n2 = 257, 5003
src = rng.standard_normal(K).astype(np.float32); idx = rng.integers(0, K, n2).astype(np.uint32)
m = (rng.integers(0, 4, n2) != 0).astype(np.uint8); val = rng.standard_normal(n2).astype(np.float32)
mem = {"src": src, "idx": idx, "mask": m, "val": val,
       "gather": R.gather(src, idx, m), "scatter_add": R.scatter(src, val, idx, m, add=True)}
for nn in (0, 1, 7, 8, 9, 1000):
    for op in ("hsum", "hprod", "hmin", "hmax"):
        mem[f"{op}_{nn}"] = np.array([R.reduce(op, (1 + 0.01 * val[:nn]).astype(np.float32) if op == "hprod" else val[:nn])], np.float32)
np.savez_compressed(os.path.join(HERE, "memory_reduce.npz"), **mem)

cfg = {}
for nn in (1000, 65536):
    A = uniform_pm1(nn, 1); X = uniform_pm1(nn, 2); B = uniform_pm1(nn, 3)
    Kc = 1024
    TA = uniform_pm1(Kc, 6); TB = uniform_pm1(Kc, 7); I = (hash_u32(np.arange(nn, dtype=np.uint64), 4) % np.uint32(Kc)).astype(np.uint32)
    cfg[f"cfg1_{nn}"] = np.array([R.cfg1(A, X, B)[0]], np.float32)
    cfg[f"cfg2_{nn}"] = np.array([R.cfg2(A, X, B)[0]], np.float32)
    y, ga, gb, _ = R.cfg3a(A, X, B); cfg[f"cfg3a_{nn}_y"] = np.array([y], np.float32); cfg[f"cfg3a_{nn}_ga"] = ga; cfg[f"cfg3a_{nn}_gb"] = gb
    y, gA, gB, _ = R.cfg3b(TA, TB, X, I); cfg[f"cfg3b_{nn}_y"] = np.array([y], np.float32); cfg[f"cfg3b_{nn}_gA"] = gA; cfg[f"cfg3b_{nn}_gB"] = gB
np.savez_compressed(os.path.join(HERE, "configs.npz"), **cfg)
print("golden fixtures written to", HERE)

# ---- BASELINE config 4 (ray-sphere, masked gather/scatter) -------------------------------------------
from test_sphere_gpu import scene, run  # noqa: E402
gx, gy, perm, mask = scene(64, seed=1)
img, h = run(R.lib.ref_cfg4, gx, gy, perm, mask)
np.savez_compressed(os.path.join(HERE, "cfg4.npz"), gx=gx, gy=gy, perm=perm, mask=mask, image=img, hits=np.array(h))

# ---- elliptic integrals (include/enoki/special.h:314-672), see oracle/ref_driver.cpp:ref_ellint_* -----------------
rng = np.random.default_rng(77)
n_e = 4096
for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
    phi = rng.uniform(-7.0, 7.0, n_e).astype(dt)            # several periods: exercises the reduction
    phi[:16] = np.linspace(-1.5, 1.5, 16).astype(dt)        # and a stretch inside the principal interval
    k = rng.uniform(-0.95, 0.95, n_e).astype(dt)
    nu = rng.uniform(-0.9, 2.0, n_e).astype(dt)
    ell = {"phi": phi, "k": k, "nu": nu, "out": R.ellint(phi, k, nu)}
    np.savez_compressed(os.path.join(HERE, f"ellint_{tag}.npz"), **ell)

# ---- Complex<FloatX>: hyperbolic and inverse functions (complex.h:196-267), see oracle/ref_driver.cpp:ref_complex_more ----
rng = np.random.default_rng(78)
ca = rng.uniform(-1.5, 1.5, (2, 2048)).astype(np.float32)
np.savez_compressed(os.path.join(HERE, "complex_more.npz"), a=ca, out=R.complex_more(ca))

# ---- transform.h (translate ... look_at), see oracle/ref_driver.cpp:ref_transform ------------------------------------
rng = np.random.default_rng(79)
tv = rng.uniform(0.3, 2.0, (3, 512)).astype(np.float32) * rng.choice([-1.0, 1.0], (3, 512)).astype(np.float32)
tp = np.stack([rng.uniform(-3, 3, 512), rng.uniform(0.4, 2.0, 512), rng.uniform(0.05, 1.0, 512), rng.uniform(5.0, 100.0, 512),
               rng.uniform(0.5, 2.0, 512), np.zeros(512)]).astype(np.float32)
np.savez_compressed(os.path.join(HERE, "transform.npz"), v=tv, p=tp, out=R.transform(tv, tp))

# ---- sh.h (real spherical harmonics, order 9), see oracle/ref_driver.cpp:ref_sh ---------------------------------------
rng = np.random.default_rng(80)
sd = rng.standard_normal((3, 1024)); sd = (sd / np.linalg.norm(sd, axis=0)).astype(np.float32)
np.savez_compressed(os.path.join(HERE, "sh.npz"), d=sd, out=R.sh(sd, 9))

