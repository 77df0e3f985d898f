"""The neighbours of the headline step (BASELINE configs[2]): the same gather -> fma -> f -> hsum -> backward() chain with another
f, a scaled loss, masked gathers, 64-bit index arrays.  Each must (a) agree with the reference build (oracle/_ref) and the float64
evaluation inside the class-D bounds and (b) stay on the bucket-ordered path -- one partition, no element-order gather, no
second count / partition in the backward sweep."""
import json

import numpy as np
import pytest

import oracle_lib as ol
from conftest import cfg3b_variant_truth, hash_u32, uniform_pm1

pytestmark = pytest.mark.gpu
N, K = 1 << 22, 1 << 20


@pytest.fixture(scope="module")
def ad():
    import enoki_amd.hip_autodiff as m
    m.hip_init(0)
    return m


@pytest.fixture(scope="module")
def ref():
    try:
        return ol.ref()
    except Exception:
        pytest.skip("oracle/_ref is not built")


@pytest.fixture(scope="module")
def data():
    A, B, x = uniform_pm1(K, 6), uniform_pm1(K, 7), uniform_pm1(N, 2)
    idx = (hash_u32(np.arange(N, dtype=np.uint64), 4) % np.uint32(K)).astype(np.uint32)
    mask = (hash_u32(np.arange(N, dtype=np.uint64), 5) & 3) != 0             # 75 % active
    return A, B, x, idx, mask


def kernels(m, fn):
    m.hip_profile_begin()
    out = fn()
    prof = json.loads(m.hip_profile_end())
    return out, {k["kernel"]: k["launches"] for k in prof if k["launches"]}


def run(ad, A, B, x, idx, mask=None, func="sin", seed=1.0, idx64=False, spelling="fmadd"):
    dA, dB = ad.Float32(A), ad.Float32(B)
    ad.set_requires_gradient(dA); ad.set_requires_gradient(dB)
    di = ad.UInt64(idx.astype(np.uint64)) if idx64 else ad.UInt32(idx)
    xd = ad.Float32(x)
    if mask is not None:
        dm = ad.Mask(mask)
        a, b = ad.gather(dA, di, dm), ad.gather(dB, di, dm)
    else:
        a, b = ad.gather(dA, di), ad.gather(dB, di)
    # BASELINE.json spells config 3b `a*x+b`: operators -- a product and a sum with a rounding each, not one fma
    u = {"fmadd": lambda: ad.fmadd(a, xd, b), "a*x+b": lambda: a * xd + b, "a*x-b": lambda: a * xd - b, "b-a*x": lambda: b - a * xd,
         "b+a*x": lambda: b + a * xd, "a*x": lambda: a * xd}[spelling]()
    y = ad.hsum(getattr(ad, func)(u))
    z = y if seed == 1.0 else y * seed
    ad.backward(z)
    gB = np.zeros(B.size, np.float32) if spelling == "a*x" else ad.gradient(dB).numpy()       # (B does not take part)
    return float(ad.detach(z).numpy()[0]), ad.gradient(dA).numpy(), gB


CASES = {
    "sin": dict(),
    "cos": dict(func="cos"),
    "exp": dict(func="exp"),
    "seed3": dict(seed=3.0),
    "exp_negative_seed": dict(func="exp", seed=-0.5),
    "masked": dict(masked=True),
    "i64": dict(idx64=True),
    "masked_exp_i64_seed": dict(func="exp", masked=True, idx64=True, seed=2.0),
    # u > 0 (the addend table shifted by 3): the derivative's factor is another function of u -- rcp(u) next to log(u), and
    # .5 / sqrt(u) = .5 rsqrt(u) next to sqrt(u) (autodiff.h:353-364) -- which the forward pass sums per table entry
    "log": dict(func="log", shift=3.0),
    "sqrt": dict(func="sqrt", shift=3.0),
    "sqrt_seed3": dict(func="sqrt", shift=3.0, seed=3.0),
    # -sqr(rcp(u)) and -.5 rsqrt(u)^3 (autodiff.h:381-403): products of unevaluated maps stay ONE map of u each
    "rcp": dict(func="rcp", shift=3.0),
    "rsqrt": dict(func="rsqrt", shift=3.0),
    # the operator spellings of u (the literal `hsum(sin(a*x+b))` of BASELINE.json configs[2] is the first): two roundings, checked
    # against the reference build evaluating the SAME spelling
    "operators": dict(spelling="a*x+b"),
    "operators_commuted": dict(spelling="b+a*x"),
    "operators_minus": dict(spelling="a*x-b", func="cos"),
    "operators_reversed_minus": dict(spelling="b-a*x", func="exp", seed=2.0),
    "operators_masked_i64": dict(spelling="a*x+b", masked=True, idx64=True),
    # ONE gather times an array (`texture lookup * weight`): the product alone stays in bucket order, the adjoint is one stream
    "product": dict(spelling="a*x"),
    "product_exp_masked_seed": dict(spelling="a*x", func="exp", masked=True, seed=-1.5),
}
# what the step may launch when it stays in bucket order: ONE partition in the forward pass, the adjoint formed there as well
EARLY = {"sin", "cos", "exp", "seed3", "exp_negative_seed", "masked", "i64", "masked_exp_i64_seed", "log", "sqrt", "sqrt_seed3", "rcp", "rsqrt",
         "operators", "operators_commuted", "operators_minus", "operators_reversed_minus", "operators_masked_i64", "product", "product_exp_masked_seed"}


@pytest.mark.parametrize("name", list(CASES))
def test_neighbour_matches_the_reference_and_stays_in_bucket_order(ad, ref, data, name):
    A, B, x, idx, mask = data
    kw = dict(CASES[name])
    m = mask if kw.pop("masked", False) else None
    idx64 = kw.pop("idx64", False)
    B = (B + np.float32(kw.pop("shift", 0.0))).astype(np.float32)
    (y, gA, gB), ks = kernels(ad, lambda: run(ad, A, B, x, idx, mask=m, idx64=idx64, **kw))
    t = cfg3b_variant_truth(A, B, x, idx, mask=m, **kw)
    ry, rgA, rgB, _ = ref.cfg3b_variant(A, B, x, idx.astype(np.uint64) if idx64 else idx, mask=m, **kw)
    assert abs(y - t["y"]) <= t["y_bound"] and abs(y - t["y"]) <= t["y_stat_bound"], (y, t["y"], t["y_stat_bound"])
    assert abs(y - ry) <= t["y_bound"] + abs(ry - t["y"])
    for g, arr, r in (("gA", gA, rgA), ("gB", gB, rgB)):
        err = np.abs(arr - t[g])
        assert np.all(err <= t[g + "_bound"]), (name, g, float((err / np.maximum(t[g + "_bound"], 1e-30)).max()))
        assert np.all(np.abs(arr - r) <= 2 * t[g + "_bound"]), (name, g)
    assert ks.get("bucket_partition") == 1, (name, ks)
    assert not any(k in ks for k in ("gather_pair_fmadd", "gather", "scatter_add_partition", "scatter_add_count", "hsum_map")), (name, ks)
    if name in EARLY:
        assert ks.get("bucket_pair_fma_reduce_adjoint") == 1 and "bucket_accumulate" not in ks, (name, ks)
        assert ks.get("scatter_add_fold") == 1, (name, ks)         # backward(): ONE fold of the per-piece tables, nothing else


@pytest.mark.parametrize("bad", [np.inf, -np.inf, np.nan])
@pytest.mark.parametrize("order", ["bucket", "element"])
def test_masked_out_lane_with_non_finite_x_is_nan_like_the_reference(ad, ref, data, bad, order):
    """A lane whose mask bit is clear gathers 0 from both tables (cuda.h:845-864): u = fma(0, x, 0), and for an infinite or NaN x
    that is NaN -- the reference's lane-by-lane hsum (dynamic.h:632-650) then is NaN.  The bucket-ordered path DROPS masked-out
    lanes in the partition; it must say NaN all the same (round 4 said f(0)), while the gradients -- the masked scatter_add drops
    the lane (cuda.h:892-905) -- are those of the finite lanes.  Checked against the reference build on the same inputs, in
    bucket order and in element order (ENOKI_HIP tuning bucket_ordered = 0)."""
    A, B, x, idx, mask = data
    x = x.copy()
    off = np.flatnonzero(~mask)
    x[off[off.size // 3]] = bad
    ry, rgA, rgB, _ = ref.cfg3b_variant(A, B, x, idx, mask=mask)
    assert np.isnan(ry), "the reference itself: masked-out lane, non-finite x -> NaN"
    if order == "element":
        ad.hip_set_tuning("bucket_ordered", 0)
    try:
        (y, gA, gB), ks = kernels(ad, lambda: run(ad, A, B, x, idx, mask=mask))
    finally:
        ad.hip_set_tuning("bucket_ordered", 1)
    assert np.isnan(y), (order, y)
    if order == "bucket":
        assert ks.get("bucket_partition") == 1 and "gather_pair_fmadd" not in ks, ks
    xf = x.copy(); xf[~mask] = 0.0            # what the masked-out lanes hold does not reach the gradients
    t = cfg3b_variant_truth(A, B, xf, idx, mask=mask)
    for g, arr, r in (("gA", gA, rgA), ("gB", gB, rgB)):
        assert np.all(np.isfinite(arr)) and np.all(np.isfinite(r)), g
        assert np.all(np.abs(arr - t[g]) <= t[g + "_bound"]), (order, g)
        assert np.all(np.abs(arr - r) <= 2 * t[g + "_bound"]), (order, g)
    # a finite x under every cleared mask bit: the masked-out lanes enter as f(0) = 0, the result is finite again
    (y2, _, _), _ = kernels(ad, lambda: run(ad, A, B, xf, idx, mask=mask))
    assert np.isfinite(y2) and abs(y2 - t["y"]) <= t["y_stat_bound"]


@pytest.mark.parametrize("where", ["middle", "all_masked"])
def test_buckets_without_elements_have_zero_gradients(ad, ref, data, where):
    """The early adjoint is summed per PIECE of a bucket and folded by backward().  A bucket that receives no element has no
    piece: its slice of both gradients must still be zero -- leading, trailing and interior empty buckets, and the degenerate
    step in which every lane is masked out."""
    A, B, x, idx, mask = data
    if where == "middle":
        idx = (idx % np.uint32(K // 4) + np.uint32(K // 2)).astype(np.uint32)           # only entries [K/2, 3K/4) are looked up
        m = None
    else:
        m = np.zeros(N, bool)
    (y, gA, gB), ks = kernels(ad, lambda: run(ad, A, B, x, idx, mask=m))
    t = cfg3b_variant_truth(A, B, x, idx, mask=m)
    ry, rgA, rgB, _ = ref.cfg3b_variant(A, B, x, idx, mask=m)
    assert abs(y - t["y"]) <= t["y_bound"] and abs(y - ry) <= t["y_bound"] + abs(ry - t["y"])
    for g, arr, r in (("gA", gA, rgA), ("gB", gB, rgB)):
        assert np.all(np.abs(arr - t[g]) <= t[g + "_bound"]), g
        assert np.all(np.abs(arr - r) <= 2 * t[g + "_bound"]), g
        untouched = t["cnt"] == 0
        assert untouched.any() and np.array_equal(arr[untouched], np.zeros(int(untouched.sum()), np.float32)), g
    if where == "middle":
        assert ks.get("bucket_pair_fma_reduce_adjoint") == 1 and ks.get("scatter_add_fold") == 1, ks
