"""PCG32 (include/enoki/random.h of the reference): integer work, so everything is bit-exact.

CPU:  oracle/enoki_oracle.c:orc_pcg32 against the unmodified reference build (PCG32<DynamicArray<Packet<float>>>) and
      against the committed fixture tests/golden/pcg32.npz.
GPU:  enoki_amd.hip.PCG32 (enoki/random.h -> ek_hip_pcg32_next, one fused kernel per draw) running the same script.
Script (oracle/ref_driver.cpp:ref_pcg32): seed(initstate, initseq[i]); `steps` masked next_uint32; next_float32;
next_uint64; next_float64; next_uint32_bounded(bound); advance(delta); next_uint32; final state."""
import os

import numpy as np
import pytest

import oracle_lib as ol
from conftest import hash_u32

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pcg32.npz")
HAVE_REF = os.path.exists(os.path.join(ol.ORACLE_DIR, "_ref", "libenoki_ref.so"))
KEYS = ["u32", "f32", "u64", "f64", "bounded", "after", "state"]
INITSTATE = 0x853c49e6748fea9b


def same(a, b):
    return all(np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)) for k in KEYS)


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref was not built (needs /root/reference)")
@pytest.mark.parametrize("n,steps,bound,delta", [(8, 1, 7, 0), (4096, 5, 1000003, 1000), (100000, 3, 2147483649, -12345),
                                                  (1024, 2, 2, 1 << 40)])
def test_oracle_pcg32_matches_reference_build(n, steps, bound, delta):
    seq = np.arange(n, dtype=np.uint64) * np.uint64(7) + np.uint64(0xda3e39cb94b95bdb)
    mask = (np.arange(n) % 3 != 0).astype(np.uint8)
    assert same(ol.port().pcg32(INITSTATE, seq, steps, mask, bound, delta), ol.ref().pcg32(INITSTATE, seq, steps, mask, bound, delta))


def test_oracle_pcg32_matches_golden():
    z = np.load(GOLDEN)
    got = ol.port().pcg32(INITSTATE, z["initseq"], 4, z["mask"], 1000003, -98765)
    assert same(got, z)
    # known answer: O'Neill's pcg32-demo, pcg32_srandom(42, 54), "Round 1" 32-bit outputs
    d = ol.port().pcg32(42, np.array([54], np.uint64), 6, np.ones(1, np.uint8), 10, 0)
    assert [hex(v) for v in d["u32"][:, 0]] == ["0xa15c02b7", "0x7b47f409", "0xba1d3330", "0x83d2f293", "0xbfa4784b", "0xcbed606e"]


def run_gpu_script(ek, initseq, steps, mask, bound, delta):
    rng = ek.PCG32(ek.UInt64(INITSTATE), ek.UInt64(initseq))
    mk = ek.Mask(mask)
    out = {"u32": np.stack([rng.next_uint32(mk).numpy() for _ in range(steps)])}
    out["f32"] = rng.next_float32().numpy()
    out["u64"] = rng.next_uint64().numpy()
    out["f64"] = rng.next_float64().numpy()
    out["bounded"] = rng.next_uint32_bounded(bound).numpy()
    rng.advance(ek.Int64(delta))
    out["after"] = rng.next_uint32().numpy()
    out["state"] = rng.state.numpy()
    return out, rng


@pytest.mark.gpu
def test_hip_pcg32_bit_exact():
    import enoki_amd.hip as ek
    z = np.load(GOLDEN)
    got, _ = run_gpu_script(ek, z["initseq"], 4, z["mask"], 1000003, -98765)
    assert same(got, z)                                              # fixture from the reference build
    for n, steps, bound, delta in [(1, 2, 10, 5), (7, 1, 3, -1), (100003, 3, 2147483649, 123456789), (1 << 20, 2, 1000, 0)]:
        seq = hash_u32(np.arange(n, dtype=np.uint64), 9).astype(np.uint64) * np.uint64(0x100000001)
        mask = ((hash_u32(np.arange(n, dtype=np.uint64), 5) & np.uint32(3)) != 0).astype(np.uint8)
        got, _ = run_gpu_script(ek, seq, steps, mask, bound, delta)
        assert same(got, ol.port().pcg32(INITSTATE, seq, steps, mask, bound, delta)), n


@pytest.mark.gpu
def test_hip_pcg32_interface():
    import enoki_amd.hip as ek
    a = ek.PCG32(ek.UInt64(42), ek.UInt64(54)); b = ek.PCG32(ek.UInt64(42), ek.UInt64(54))
    assert a == b and hex(int(a.next_uint32().numpy()[0])) == "0xa15c02b7"      # scalar generator, pcg32-demo known answer
    assert ek.PCG32() == ek.PCG32()
    assert a != b
    for _ in range(9):
        a.next_uint32()
    assert int((a - b).numpy()[0]) == 10 and int((b - a).numpy()[0]) == -10
    b.advance(ek.Int64(10))
    assert a == b
    n = 1 << 20
    rng = ek.PCG32(ek.UInt64(INITSTATE), ek.UInt64.arange(n))
    f = rng.next_float32().numpy(); d = rng.next_float64().numpy()
    assert f.min() >= 0 and f.max() < 1 and abs(f.mean() - 0.5) < 2e-3 and d.min() >= 0 and d.max() < 1 and abs(d.mean() - 0.5) < 2e-3
    r = rng.next_uint64_bounded(1000).numpy()
    assert r.max() < 1000 and abs(r.mean() - 499.5) < 2
