import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_collection_modifyitems(config, items):
    """the `extras` suites run only when asked for by name: they are neither part of the GPU gate (-m gpu) nor runnable on CPU"""
    if "extras" in (config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="support-library device test outside SURVEY section 8: run with -m extras on a GPU box")
    for item in items:
        if "extras" in item.keywords:
            item.add_marker(skip)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "extras: device tests of the support library OUTSIDE the hot-path contract of SURVEY section 8 "
                                       "(complex, matrix, morton, sh, special); need a GPU, run with `-m extras`")
    # torch ships its own HIP runtime; when both runtimes live in one process torch's must initialise first
    # (bench.py does the same), otherwise torch later reports "No HIP GPUs are available".
    if "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or ""):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass


def hash_u32(i, seed):
    """counter-based integer hash shared by host and device generators"""
    v = (np.asarray(i, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF)
    v = v.astype(np.uint32)
    v ^= v >> np.uint32(16)
    v = (v.astype(np.uint64) * np.uint64(0x7FEB352D) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    v ^= v >> np.uint32(15)
    v = (v.astype(np.uint64) * np.uint64(0x846CA68B) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    v ^= v >> np.uint32(16)
    return v


def uniform_pm1(n, seed):
    """f32 uniform in [-1, 1): 2u - 1 with u = (h >> 8) * 2^-24 (exact in f32)"""
    h = hash_u32(np.arange(n, dtype=np.uint64), seed)
    u = (h >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    return (np.float32(2.0) * u - np.float32(1.0)).astype(np.float32)


SPECIALS_F32 = np.array(
    [0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1e-38, 1.17549435e-38, 3.4e38, -3.4e38, 1, -1, 0.5, 2,
     88.3, 88.5, -88.3, -88.5, -103, 8192, -8192, 1e6, 1e9, 3e9, -3e9, 1e20, 0.70710678, 0.7071068, np.pi,
     np.pi / 2, np.pi / 4, 1.5, 2.5, -1.5, -2.5, 0.49999997, 8388608.0, 16777216.0], dtype=np.float32)


def f32_inputs(n, seed, scale=1.0, specials=True):
    rng = np.random.default_rng(seed)
    v = (rng.standard_normal(n) * scale).astype(np.float32)
    if specials and n >= SPECIALS_F32.size:
        v[:SPECIALS_F32.size] = SPECIALS_F32
    return v


SPECIALS_F64 = np.array(
    [0.0, -0.0, np.inf, -np.inf, np.nan, 5e-324, -5e-324, 1e-310, 2.2250738585072014e-308, 1e308, -1e308, 1, -1, 0.5, 2,
     709.4, 709.5, -709.4, -709.5, -745.2, 8192, -8192, 1e6, 1e9, 3e9, -3e9, 0.70710678118654752, 0.7071067811865476,
     np.pi, np.pi / 2, np.pi / 4, 1.5, 2.5, -1.5, -2.5, 0.49999999999999994, 4503599627370496.0, 0.125, 8.0, 3.0],
    dtype=np.float64)


def f64_inputs(n, seed, scale=1.0, specials=True, limit=None):
    """float64 test inputs; `limit` clamps magnitudes (the AVX2 reference build's f64 sin/cos are indeterminate
    for |x| * 4/pi >= 2^32: its 64-bit integer packets convert through 32-bit lanes)"""
    rng = np.random.default_rng(seed)
    v = rng.standard_normal(n) * scale
    if specials and n >= SPECIALS_F64.size:
        v[:SPECIALS_F64.size] = SPECIALS_F64
    if limit is not None:
        big = np.abs(v) > limit
        v[big & np.isfinite(v)] = limit
    return v


def bits_equal(a, b):
    """bit equality, except that any NaN equals any NaN (payloads are not part of the contract)"""
    a = np.asarray(a); b = np.asarray(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    if a.dtype.kind == "f":
        ia = a.view(np.uint32 if a.dtype == np.float32 else np.uint64)
        ib = b.view(np.uint32 if b.dtype == np.float32 else np.uint64)
        return bool(np.all((ia == ib) | (np.isnan(a) & np.isnan(b))))
    return bool(np.array_equal(a, b))


def ulp_diff(a, b):
    """distance in units of the last place between two f32 arrays (finite entries)"""
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    ia = a.view(np.int32).astype(np.int64); ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, np.int64(-2147483648) - ia, ia)
    ib = np.where(ib < 0, np.int64(-2147483648) - ib, ib)
    return np.abs(ia - ib)


# ---- class C: functions that go through rcp() / rsqrt() ------------------------------------------------------------
# The reference's own rows do not agree on them: the AVX2 row approximates reciprocals (rcpps + one Newton step,
# array_avx.h:324-395) but fuses its polynomial cores; the scalar `none` row divides exactly (array_fallbacks.h:23-101) but its
# generic packets never fuse (array_static.h:433-446).  The device divides exactly AND fuses -- exactly what an FMA machine
# does with a correctly rounded division.  What is asserted (tests/golden/classc_scalar.npz, made from both rows):
#   * rcp, rsqrt, division: BIT-EXACT against the scalar row;
#   * everything built on them: no further from EITHER row than the two rows are from each other, and within the listed
#     number of ulp (the worst case observed over the fixture; the share of differing elements is in DESIGN section 5).
CLASS_C_EXACT = ["rcp", "rsqrt"]
#                op: (max ulp vs scalar row, max ulp vs AVX2 row, max ulp between the two rows of the reference)
CLASS_C_BAND = {"tan": (2, 2, 3), "cot": (2, 2, 3), "sinh": (3, 3, 3), "cosh": (2, 3, 3), "tanh": (2, 4, 4),
                "erf": (2, 2, 2), "erfc": (8, 4, 8), "i0e": (5, 4, 5)}


def class_c_fixture():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "classc_scalar.npz"))


def class_c_check(op, got, z):
    """`got` = op(x) of the implementation under test on the fixture's inputs (see make_golden.py for the argument)"""
    s, a = z[f"scalar_{op}"], z[f"avx2_{op}"]
    if op in CLASS_C_EXACT:
        assert bits_equal(got, s), op
        return
    ok = np.isfinite(got) & np.isfinite(s) & np.isfinite(a) & (np.abs(got) > 1e-30)
    assert ok.mean() > 0.9, op
    # outside `ok` (overflow, underflow to denormals / zero, NaN): same class of result as the scalar row
    assert np.array_equal(np.isnan(got), np.isnan(s)) and np.array_equal(np.isinf(got), np.isinf(s)), op
    ds, da, dr = ulp_diff(got[ok], s[ok]).max(), ulp_diff(got[ok], a[ok]).max(), ulp_diff(s[ok], a[ok]).max()
    bs, ba, br = CLASS_C_BAND[op]
    assert dr == br, (op, "the fixture's own rows", dr)
    assert ds <= bs and da <= ba and ds <= dr and da <= dr, (op, ds, da, dr)


def class_c_arg(op, z):
    return np.abs(z["x"]) + np.float32(1e-3) if op == "rsqrt" else z["x"]


def hsum_depth(n):
    """longest chain of fp additions behind one output of the library's hsum (csrc/reduce.hip): 2^20 lane accumulators in
    stage 1, each summing ceil(n / 2^20) entries in sequence, then the wave / workgroup trees and stage 2"""
    return -(-n // (1 << 20)) + 10 + 14


def cfg3b_truth(A, B, x, idx):
    """float64 evaluation of BASELINE config 3b (y = hsum(sin(A[idx] x + B[idx])), gradients w.r.t. A and B) with the
    class-D bounds of SURVEY 8c for OUR summation orders.  A sum of m terms evaluated in any order of depth d is within
    d * 2^-24 * sum|terms| of the exact sum of the ROUNDED terms; the terms themselves (u = fma(a, x, b) in f32, then
    sin / cos of it, then a product) are within 4 * 2^-24 ABSOLUTE of their exact values for |a|, |x|, |b| <= 1 (the
    rounding of u moves sin u by at most 2 * 2^-24, not by a relative amount)."""
    eps = 2.0 ** -24
    K, n = A.size, x.size
    ii = idx.astype(np.int64)
    x64 = x.astype(np.float64)
    u = A.astype(np.float64)[ii] * x64 + B.astype(np.float64)[ii]
    s, c = np.sin(u), np.cos(u)
    cnt = np.bincount(ii, minlength=K)
    sum_abs = float(np.abs(s).sum())
    out = {"y": float(s.sum()), "y_bound": eps * (hsum_depth(n) * sum_abs + 4 * n), "cnt": cnt,
           "y_stat_bound": stat_sum_bound(s, hsum_depth(n)),
           # the reference adds lane-wise: 8 (AVX2) accumulators of n / 8 entries each, then a tree (dynamic.h:632-650)
           "y_bound_reference": eps * ((n // 8 + 4) * sum_abs + 4 * n)}
    for name, terms in (("gA", c * x64), ("gB", c)):
        out[name] = np.bincount(ii, weights=terms, minlength=K)
        out[name + "_bound"] = eps * (cnt * np.bincount(ii, weights=np.abs(terms), minlength=K) + 4 * cnt)
    return out


def cfg3b_variant_truth(A, B, x, idx, mask=None, func="sin", seed=1.0, spelling="fmadd"):
    """cfg3b_truth for the neighbours that bench.py times next to the headline: y = seed * hsum(f(u)), f = sin | cos | exp | log | sqrt,
    masked-out lanes gather 0 (u = 0, no gradient).  Same class-D bounds, scaled by |seed| and by the size of f and f'."""
    eps = 2.0 ** -24
    K, n = A.size, x.size
    ii = idx.astype(np.int64)
    on = np.ones(n, bool) if mask is None else np.asarray(mask, bool)
    x64 = x.astype(np.float64)
    # (the operator spellings differ from fmadd by one more rounding of u, which the bounds below cover: |f'| <= big)
    sa, sb = {"fmadd": (1, 1), "a*x+b": (1, 1), "b+a*x": (1, 1), "a*x-b": (1, -1), "b-a*x": (-1, 1), "a*x": (1, 0)}[spelling]
    u = np.where(on, sa * A.astype(np.float64)[ii] * x64 + sb * B.astype(np.float64)[ii], 0.0)
    with np.errstate(all="ignore"):         # (log / sqrt want positive u: the tests that use them shift the addend table)
        safe = np.where(on, u, 1.0)
    f, df = {"sin": (np.sin, np.cos), "cos": (np.cos, lambda v: -np.sin(v)), "exp": (np.exp, np.exp),
             "log": (lambda v: np.where(on, np.log(safe), 0.0), lambda v: np.where(on, 1.0 / safe, 0.0)),
             "sqrt": (lambda v: np.where(on, np.sqrt(safe), 0.0), lambda v: np.where(on, 0.5 / np.sqrt(safe), 0.0)),
             "rcp": (lambda v: np.where(on, 1.0 / safe, 0.0), lambda v: np.where(on, -1.0 / (safe * safe), 0.0)),
             "rsqrt": (lambda v: np.where(on, safe ** -0.5, 0.0), lambda v: np.where(on, -0.5 * safe ** -1.5, 0.0))}[func]
    s, c = f(u) * seed, df(u) * seed
    big = max(1.0, float(np.abs(s).max()), float(np.abs(c).max()))          # |f|, |f'| <= e^2 for exp on |u| <= 2
    cnt = np.bincount(ii[on], minlength=K)
    out = {"y": float(s.sum()), "y_bound": eps * (hsum_depth(n) * float(np.abs(s).sum()) + 8 * big * n), "cnt": cnt,
           "y_stat_bound": stat_sum_bound(s, hsum_depth(n)) + 4 * eps * abs(float(s.sum()))}
    for name, terms in (("gA", sa * c * x64), ("gB", sb * c)):
        out[name] = np.bincount(ii[on], weights=terms[on], minlength=K)
        out[name + "_bound"] = eps * (cnt * np.bincount(ii[on], weights=np.abs(terms[on]), minlength=K) + 8 * big * cnt)
    return out


def stat_sum_bound(terms64, depth, sigmas=5.0):
    """What the error of an f32 sum of these terms looks like when roundings behave like independent noise (they do): for
    sequential chains of `depth` additions followed by a balanced tree, sigma^2 ~= u^2 (depth / 6 + 8) sum t^2 from the
    zero-mean part of the partial sums (chains: u^2 r^2 k / 3 per addition; tree: u^2 r^2 n / 3 per level, ~23 levels), another
    u^2 * 4 sum t^2 for the rounding of the f32 terms themselves, plus the drift when the terms have a mean m: 0.8 u |sum t|
    from the top of the tree and u |m| depth sqrt(n) / 3 from the chains.  Returns `sigmas` standard deviations -- for
    BASELINE config 3b at 64 Mi elements 7e-3, against 1.2e-3 observed and a worst-case bound of 188."""
    t = np.asarray(terms64, np.float64)
    u, n, total = 2.0 ** -24, t.size, float(t.sum())
    sigma = u * np.sqrt((depth / 6.0 + 12.0) * float((t * t).sum()))
    drift = u * (0.8 * abs(total) + abs(total) / max(n, 1) * depth * np.sqrt(n) / 3.0)
    return sigmas * (sigma + drift)


def hsum_bound(terms64, per_term_ulps=4):
    """class-D bound of the library's hsum over float32 roundings of `terms64` (see cfg3b_truth)"""
    n = terms64.size
    return 2.0 ** -24 * (hsum_depth(n) * float(np.abs(terms64).sum()) + per_term_ulps * n)


@pytest.fixture(scope="session")
def capi():
    from enoki_amd import capi as c
    c.init()
    return c


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.port()
