"""round 5: what the page partition's class weights (ek_paged.h) settle at on this box, and the classes' loop durations with and
without them.  python tools/probe_xcd_balance.py   (GPU box)"""
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import enoki_amd.hip as ekc
import enoki_amd.hip_autodiff as ek
from enoki_amd import capi, synth

ek.hip_init(0)
n, K = 1 << 26, 1 << 20
A0, B0 = synth.uniform_pm1(0, K, 6), synth.uniform_pm1(0, K, 7)
x = ek.Float32(synth.uniform_pm1(0, n, 2))
idx = ek.UInt32(synth.hash_u32(0, n, 4) % ekc.UInt32(K))


def step():
    A, B = ek.Float32(A0), ek.Float32(B0)
    ek.set_requires_gradient(A); ek.set_requires_gradient(B)
    y = ek.hsum(ek.sin(ek.fmadd(ek.gather(A, idx), x, ek.gather(B, idx))))
    ek.backward(y)
    return ek.gradient(A), ek.gradient(B)


def state():
    w, t, d = (ctypes.c_uint32 * 8)(), (ctypes.c_uint32 * 8)(), ctypes.c_uint32()
    capi.check(capi.lib.ek_hip_partition_class_state(w, t, ctypes.byref(d)))
    return [round(v / 65536, 4) for v in w] + ["dealt" if d.value else "equal"], [round(v / 100, 1) for v in t]


def kernel_us(reps=40):
    ek.hip_profile_begin()
    for _ in range(reps):
        step()
    prof = json.loads(ek.hip_profile_end())
    return {k["kernel"]: round(k["total_ms"] / k["launches"] * 1e3, 1) for k in prof if k["launches"] >= reps and k["total_ms"] / k["launches"] > 0.004}


for balance in (1, 2, 1, 2):          # 2: equal chunks, loops stamped
    ek.hip_set_tuning("xcd_balance", balance)
    for _ in range(300):
        step()
    w, t = state()
    print("xcd_balance", balance, "weights", w, "loop us by class", t, "spread %.1f us" % (max(t) - min(t)), kernel_us(), flush=True)
