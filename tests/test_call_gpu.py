"""Vectorised virtual method calls on device pointer arrays (include/enoki/array_call.h; reference
include/enoki/array_call.h:17-283, cuda.h:815-842, horiz.cu:35-122, tests/call.cpp, tests/autodiff.cpp:564-607).

tests/cpp/call_hip.cpp holds a two-class hierarchy; every lane's expected value is recomputed here with the CPU
oracle's elementwise ops (all class A), so the comparison is bit-exact; partition() is checked against numpy."""
import ctypes
import os

import numpy as np
import pytest

from conftest import bits_equal, hash_u32, uniform_pm1

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def run(which, x, t, mask):
    lib = ctypes.CDLL(os.path.join(HERE, "cpp", "libcall_hip.so"))
    n = x.size
    o = {"eval": np.empty(n, np.float32), "eval_masked": np.empty(n, np.float32), "off": np.empty((3, n), np.float32),
         "id": np.empty(n, np.float32), "touch": np.zeros(6, np.uint64), "d": np.empty(n, np.float32),
         "grad": np.empty(n, np.float32), "groups": np.zeros(9, np.uint32), "perm": np.zeros(n, np.uint32)}
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = lib.hip_call_test(p(which), p(x), p(t), p(mask), ctypes.c_size_t(n), p(o["eval"]), p(o["eval_masked"]), p(o["off"]),
                           p(o["id"]), p(o["touch"]), p(o["d"]), p(o["grad"]), p(o["groups"]), p(o["perm"]))
    assert rc == 0, rc
    return o


def expected(oracle, which, x, t, mask, single):
    f32 = np.float32
    sinx = oracle.unary("sin", x)
    f = {0: sinx * f32(1.5), 1: oracle.ternary("fmadd", x, np.full_like(x, -2.0), np.full_like(x, 0.25)), 2: sinx * f32(-0.5)}
    inactive = {0: f32(-1), 1: f32(-2), 2: f32(-1)}
    ev = np.zeros_like(x); evm = np.zeros_like(x); off = np.zeros((3, x.size), f32); ident = np.zeros_like(x)
    d = np.zeros_like(x)
    for k in (0, 1, 2):
        sel = which == k
        ev[sel] = f[k][sel]
        act = sel & ((mask != 0) | single)          # a single-instance array is called with mask = true (array_call.h:151-153)
        evm[sel] = inactive[k]
        evm[act] = f[k][act]
        ident[sel] = 1.0 if k != 1 else 2.0
        amp = {0: f32(1.5), 2: f32(-0.5)}.get(k)
        if k == 1:
            off[0][sel] = (x * t)[sel]; off[1][sel] = (t * t)[sel]; off[2][sel] = t[sel]
            d[sel] = (f[1] * x)[sel]
        else:
            off[0][sel] = (x + t)[sel]; off[1][sel] = (t + f32(0))[sel]; off[2][sel] = f32(1) + amp
            d[sel] = f[k][sel]
    return ev, evm, off, ident, d


@pytest.mark.parametrize("n", [5, 1000, 100003])
def test_vectorised_calls_bit_exact(oracle, n):
    which = (hash_u32(np.arange(n, dtype=np.uint64), 11) % np.uint32(4)).astype(np.uint8)
    which[which == 3] = 255                                   # null pointers
    x = uniform_pm1(n, 12) * np.float32(3); t = uniform_pm1(n, 13)
    mask = ((hash_u32(np.arange(n, dtype=np.uint64), 5) & np.uint32(3)) != 0).astype(np.uint8)
    o = run(which, x, t, mask)
    ev, evm, off, ident, d = expected(oracle, which, x, t, mask, single=False)
    assert bits_equal(o["eval"], ev) and bits_equal(o["eval_masked"], evm)
    assert bits_equal(o["off"], off) and bits_equal(o["id"], ident) and bits_equal(o["d"], d)
    # derivatives: cos(x) * amp resp. d/dx (c1 x + c0) x
    cosx = np.cos(x.astype(np.float64))
    g = np.where(which == 0, 1.5 * cosx, np.where(which == 2, -0.5 * cosx, np.where(which == 1, -4.0 * x + 0.25, 0.0)))
    assert np.allclose(o["grad"], g, rtol=2e-6, atol=2e-6)
    # side effects: every instance saw its own lanes once, and the mask restricted to them
    for k in (0, 1, 2):
        assert o["touch"][2 * k] == (which == k).sum() and o["touch"][2 * k + 1] == ((which == k) & (mask != 0)).sum()
    # partition(): one group per distinct pointer (incl. null), ascending pointer order is asserted inside the library;
    # lanes ascending within a group (stable), sizes = histogram
    ng = int(o["groups"][0]); pos = 0
    assert ng == len(np.unique(which))
    seen = set()
    for gidx in range(ng):
        number, size = int(o["groups"][1 + 2 * gidx]), int(o["groups"][2 + 2 * gidx])
        lanes = o["perm"][pos:pos + size]; pos += size
        assert np.array_equal(lanes, np.flatnonzero(which == number).astype(np.uint32)), number
        seen.add(number)
    assert seen == set(np.unique(which).tolist()) and pos == n


def test_single_instance_shortcut(oracle):
    n = 4099
    which = np.zeros(n, np.uint8)
    x = uniform_pm1(n, 21) * np.float32(3); t = uniform_pm1(n, 22)
    mask = (np.arange(n) % 2).astype(np.uint8)
    o = run(which, x, t, mask)
    ev, evm, off, ident, d = expected(oracle, which, x, t, mask, single=True)
    assert bits_equal(o["eval"], ev) and bits_equal(o["eval_masked"], evm) and bits_equal(o["off"], off)
    assert bits_equal(o["id"], ident) and bits_equal(o["d"], d)
    assert int(o["groups"][0]) == 1 and int(o["groups"][2]) == n and o["touch"][0] == n


def test_struct_arguments_and_structwise_memory_ops():
    """ENOKI_STRUCT / ENOKI_STRUCT_SUPPORT (array_macro.h:216-359): a struct of arrays as argument and result of a
    vectorised call, plus struct-wise gather / scatter / zero / slices / select"""
    lib = ctypes.CDLL(os.path.join(HERE, "cpp", "libcall_hip.so"))
    n = 10007
    which = (hash_u32(np.arange(n, dtype=np.uint64), 31) % np.uint32(3)).astype(np.uint8)
    which[which == 2] = 255
    x = uniform_pm1(n, 32) * np.float32(4); dt = uniform_pm1(n, 33)
    f32 = np.float32
    o = {k: np.empty(n, np.float32) for k in ("px", "py", "pz", "t", "gx", "gt")}
    valid = np.empty(n, np.uint8); zs = np.zeros(2, np.uint64)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = lib.hip_struct_test(p(which), p(x), p(dt), ctypes.c_size_t(n), p(o["px"]), p(o["py"]), p(o["pz"]), p(o["t"]), p(valid),
                             p(o["gx"]), p(o["gt"]), p(zs))
    assert rc == 0, rc
    fwd, flip = which == 0, which == 1
    px = np.where(fwd, x + dt, np.where(flip, -x, f32(0))).astype(f32)
    py = np.where(fwd, x * f32(2) + f32(0), np.where(flip, -(x * f32(2)), f32(0))).astype(f32)
    pz = np.where(fwd, f32(1), np.where(flip, f32(-1), f32(0))).astype(f32)
    t = np.where(fwd, (x + f32(1)) + dt, np.where(flip, (x + f32(1)) * dt, f32(0))).astype(f32)
    v = np.where(fwd, x > 0, np.where(flip, ~(x > 0), False))
    assert bits_equal(o["px"], px) and bits_equal(o["py"], py) and bits_equal(o["pz"], pz) and bits_equal(o["t"], t)
    assert np.array_equal(valid != 0, v)
    assert bits_equal(o["gx"], px[::-1].copy()) and bits_equal(o["gt"], t[::-1].copy())
    assert zs[0] == n and zs[1] == n


def test_masked_assignment():
    """masked(x, m) = v / x[m] op= v (array_masked.h; selects on dynamic arrays) for arrays, nested arrays, structs,
    differentiable arrays and plain scalars"""
    lib = ctypes.CDLL(os.path.join(HERE, "cpp", "libcall_hip.so"))
    n = 10007
    x = uniform_pm1(n, 41) * np.float32(2)
    o = {k: np.empty(n, np.float32) for k in ("assign", "add", "vec_y", "struct_t", "grad")}
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = lib.hip_masked_test(p(x), ctypes.c_size_t(n), p(o["assign"]), p(o["add"]), p(o["vec_y"]), p(o["struct_t"]), p(o["grad"]))
    assert rc == 0, rc
    f32 = np.float32
    pos = x > 0
    assert bits_equal(o["assign"], np.where(pos, f32(5), x))
    assert bits_equal(o["add"], np.where(pos, x + x * f32(2), x * f32(-1)))
    assert bits_equal(o["vec_y"], np.where(pos, f32(7), x + f32(1)))
    assert bits_equal(o["struct_t"], np.where(pos, x, x * f32(0) - f32(3)))
    assert np.allclose(o["grad"], np.where(pos, 3.0, 2.0 * x), rtol=1e-6)


def test_call_support_getters():
    """ENOKI_CALL_SUPPORT_GETTER: per-lane values of scalar data members (null pointer / masked lanes -> 0)"""
    lib = ctypes.CDLL(os.path.join(HERE, "cpp", "libcall_hip.so"))
    n = 5003
    which = (hash_u32(np.arange(n, dtype=np.uint64), 51) % np.uint32(4)).astype(np.uint8)
    which[which == 3] = 255
    mask = (np.arange(n) % 3 != 0).astype(np.uint8)
    tag = np.empty(n, np.float32); tagm = np.empty(n, np.float32); lanes = np.empty(n, np.uint64)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert lib.hip_getter_test(p(which), p(mask), ctypes.c_size_t(n), p(tag), p(tagm), p(lanes)) == 0
    want = np.select([which == 0, which == 1, which == 2], [10.5, -3.25, 7.0], 0.0).astype(np.float32)
    assert np.array_equal(tag, want) and np.array_equal(tagm, np.where(mask != 0, want, np.float32(0)))
    assert np.array_equal(lanes, np.select([which == 0, which == 1, which == 2], [11, 22, 33], 0).astype(np.uint64))


@pytest.mark.parametrize("instances", [1, 3, 5000])
def test_call_support_getters_on_the_device(instances):
    """instances of a class with ENOKI_PINNED_OPERATOR_NEW live where the GPU can read them: the getter is ONE gather out of
    instance memory (ek_hip_gather_address; array_call.h:269-283 does the same from managed memory) however many instances
    there are -- no partition, no host read per instance"""
    lib = ctypes.CDLL(os.path.join(HERE, "cpp", "libcall_hip.so"))
    n = 200003
    rng = np.random.default_rng(instances)
    which = rng.integers(0, instances + 1, n).astype(np.uint32)
    which[which == instances] = 0xFFFFFFFF
    mask = (np.arange(n) % 3 != 0).astype(np.uint8)
    power = np.empty(n, np.float32); tag = np.empty(n, np.float32); samples = np.empty(n, np.uint32); launches = np.zeros(1, np.uint64)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert lib.hip_getter_device_test(p(which), p(mask), ctypes.c_size_t(n), ctypes.c_uint32(instances), p(power), p(tag), p(samples),
                                      p(launches)) == 0
    live = which != 0xFFFFFFFF
    k = np.where(live, which, 0).astype(np.float64)
    assert np.array_equal(power, np.where(live, 0.5 * k + 1.0, 0.0).astype(np.float32))
    assert np.array_equal(tag, np.where(live & (mask != 0), 1000.0 - k, 0.0).astype(np.float32))
    assert np.array_equal(samples, np.where(live, 7 * k + 3, 0).astype(np.uint32))
    assert int(launches[0]) <= 5, launches          # three gathers + the double -> float cast (+ a mask and): not per instance


@pytest.mark.parametrize("instances", [1, 3, 1000, 70000])
def test_partition_scales_with_instances(instances):
    """partition() = dense 32-bit keys + ONE stable radix sort + run starts (like cuda_partition, horiz.cu:35-122): groups
    in ascending pointer order, lanes ascending inside a group, null pointers first; the number of kernel launches does
    not depend on the number of distinct instances"""
    lib = ctypes.CDLL(os.path.join(HERE, "cpp", "libcall_hip.so"))
    n = 1 << 20
    rng = np.random.default_rng(instances)
    which = rng.integers(0, instances, n).astype(np.uint32)
    which[rng.integers(0, n, n // 50)] = 0xFFFFFFFF               # some null pointers
    gi = np.zeros(instances + 1, np.uint32); gs = np.zeros(instances + 1, np.uint32); ng = ctypes.c_uint32()
    perm = np.zeros(n, np.uint32); ms = ctypes.c_double(); launches = ctypes.c_uint64()
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert lib.hip_partition_many(p(which), ctypes.c_size_t(n), ctypes.c_uint32(instances), p(gi), p(gs), ctypes.byref(ng), p(perm),
                                  ctypes.byref(ms), ctypes.byref(launches)) == 0
    g = ng.value
    present = np.unique(which)
    expect_order = np.concatenate([[0xFFFFFFFF], present[present != 0xFFFFFFFF]]).astype(np.uint32)     # null (0) sorts first
    assert g == expect_order.size and np.array_equal(gi[:g], expect_order)
    at = 0
    for k in range(g):
        lanes = perm[at:at + gs[k]]
        assert np.array_equal(lanes, np.flatnonzero(which == gi[k]).astype(np.uint32)), k
        at += gs[k]
    assert at == n
    assert launches.value <= 40, launches.value                    # independent of the number of instances
    print(f"partition of {n} lanes over {instances} instances: {ms.value:.3f} ms, {launches.value} launches")
