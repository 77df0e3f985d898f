"""The C-ABI exchange step (ek_hip_dist_*, csrc/dist.cpp): RCCL on the library stream for callers without python / torch.
One GPU: a world of one rank through a REAL communicator (ncclCommInitRank with nranks = 1) and through the RCCL-free path;
two GPUs: two processes, the unique id shipped through a file (skipped on a 1-GPU box)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SZ = ctypes.c_size_t


def test_shard_ranges_partition_every_length():
    """CPU part: the index ranges of all ranks tile [0, n) exactly (the same partition for every size-n array)"""
    from enoki_amd import capi
    for n in (0, 1, 7, 1 << 20, (1 << 26) + 3):
        for world in (1, 2, 3, 8):
            edges = []
            for r in range(world):
                b, e = SZ(), SZ()
                assert capi.lib.ek_hip_dist_shard_range(SZ(n), r, world, ctypes.byref(b), ctypes.byref(e)) == 0
                edges.append((b.value, e.value))
            assert edges[0][0] == 0 and edges[-1][1] == n and all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            assert max(e - b for b, e in edges) - min(e - b for b, e in edges) <= 1


@pytest.mark.gpu
@pytest.mark.parametrize("with_rccl", [False, True])
def test_world_of_one(with_rccl):
    from enoki_amd import capi
    capi.init()
    lib = capi.lib
    ident = (ctypes.c_char * 128)()
    if with_rccl:
        capi.check(lib.ek_hip_dist_unique_id(ident))
    capi.check(lib.ek_hip_dist_init(0, 1, ident if with_rccl else None))
    try:
        a = np.arange(1000, dtype=np.float32)
        buf = capi.Buf.from_numpy(a)
        capi.check(lib.ek_hip_dist_all_reduce(buf.ek, 0, ctypes.c_void_p(buf.ptr), SZ(buf.n)))
        out = capi.Buf(np.float32, 1000)
        capi.check(lib.ek_hip_dist_reduce_scatter(buf.ek, 0, ctypes.c_void_p(out.ptr), ctypes.c_void_p(buf.ptr), SZ(1000)))
        gathered = capi.Buf(np.float32, 1000)
        capi.check(lib.ek_hip_dist_all_gather(buf.ek, ctypes.c_void_p(gathered.ptr), ctypes.c_void_p(out.ptr), SZ(1000)))
        capi.sync()
        assert np.array_equal(buf.numpy(), a) and np.array_equal(out.numpy(), a) and np.array_equal(gathered.numpy(), a)
        r, w = ctypes.c_int(), ctypes.c_int()
        lib.ek_hip_dist_world(ctypes.byref(r), ctypes.byref(w))
        assert (r.value, w.value) == (0, 1)
    finally:
        capi.check(lib.ek_hip_dist_finalize())


@pytest.mark.gpu
def test_collective_under_capture_is_unsupported_and_the_capture_stays_valid():
    """include/enoki_hip.h: the RCCL collectives refuse to be recorded into a step graph with EK_ERR_UNSUPPORTED -- the code a caller
    treats as `end the capture, run this step eagerly` -- not with a hard EK_ERR_INVALID"""
    from enoki_amd import capi
    capi.init()
    lib = capi.lib
    ident = (ctypes.c_char * 128)()
    capi.check(lib.ek_hip_dist_unique_id(ident))
    capi.check(lib.ek_hip_dist_init(0, 1, ident))            # (a real communicator: without one a world of one is a no-op, capture or not)
    try:
        a = np.arange(4096, dtype=np.float32)
        buf = capi.Buf.from_numpy(a)
        out = capi.Buf(np.float32, 4096)
        capi.check(lib.ek_hip_graph_begin())
        try:
            rc = lib.ek_hip_dist_all_reduce(buf.ek, 0, ctypes.c_void_p(buf.ptr), SZ(buf.n))
            assert rc == -2, f"EK_ERR_UNSUPPORTED expected under capture, got {rc}"
            assert lib.ek_hip_dist_reduce_scatter(buf.ek, 0, ctypes.c_void_p(out.ptr), ctypes.c_void_p(buf.ptr), SZ(4096)) == -2
            assert lib.ek_hip_dist_all_gather(buf.ek, ctypes.c_void_p(out.ptr), ctypes.c_void_p(buf.ptr), SZ(4096)) == -2
        finally:
            g = ctypes.c_void_p()
            capi.check(lib.ek_hip_graph_end(ctypes.byref(g)))          # the capture itself is intact
        if g.value:
            capi.check(lib.ek_hip_graph_destroy(g))
        capi.check(lib.ek_hip_dist_all_reduce(buf.ek, 0, ctypes.c_void_p(buf.ptr), SZ(buf.n)))        # and eagerly it runs
        capi.sync()
        assert np.array_equal(buf.numpy(), a)
    finally:
        capi.check(lib.ek_hip_dist_finalize())


@pytest.mark.gpu
def test_one_rccl_copy_per_process():
    """csrc/dist.cpp loads RCCL at run time.  In a python process torch has its own librccl.so mapped (the copy torch.distributed
    talks to): the C-ABI exchange must reuse THAT copy instead of mapping /opt/rocm/lib/librccl.so next to it -- two RCCL runtimes
    in one process have separate bootstrap state.  Checked in a fresh interpreter: torch first, then a 1-rank communicator
    through the C ABI; exactly one librccl object may be mapped, and it is the one ek_hip_dist_rccl_path() names."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import ctypes, os, sys
sys.path.insert(0, sys.argv[1])
import torch
torch.cuda.init()
from enoki_amd import capi
capi.init()
lib = capi.lib
lib.ek_hip_dist_rccl_path.restype = ctypes.c_char_p
ident = (ctypes.c_char * 128)()
capi.check(lib.ek_hip_dist_unique_id(ident))
capi.check(lib.ek_hip_dist_init(0, 1, ident))
import numpy as np
buf = capi.Buf.from_numpy(np.arange(64, dtype=np.float32))
capi.check(lib.ek_hip_dist_all_reduce(buf.ek, 0, ctypes.c_void_p(buf.ptr), ctypes.c_size_t(64)))
capi.check(lib.ek_hip_dist_all_reduce(buf.ek, 0, None, ctypes.c_size_t(0)))        # n == 0: a no-op, not an error
capi.sync()
assert np.array_equal(buf.numpy(), np.arange(64, dtype=np.float32))
capi.check(lib.ek_hip_dist_finalize())
mapped = sorted({l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l})
print("EK_RCCL_COPY=" + lib.ek_hip_dist_rccl_path().decode() + " | " + ";".join(mapped))
'''
    out = subprocess.run([sys.executable, "-c", code, root], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("EK_RCCL_COPY=")][-1]
    used, mapped = line[len("EK_RCCL_COPY="):].split(" | ")
    mapped = [m for m in mapped.split(";") if m]
    assert len(mapped) == 1, f"more than one RCCL copy mapped: {mapped}"
    assert os.path.realpath(used) == os.path.realpath(mapped[0]), (used, mapped)
    assert os.sep + "torch" + os.sep in os.path.realpath(used), f"expected torch's own copy, got {used}"


WORKER = r"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
rank, world, path = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
os.environ["HIP_VISIBLE_DEVICES"] = str(rank)
from enoki_amd import capi
capi.init(); lib = capi.lib; SZ = ctypes.c_size_t
ident = (ctypes.c_char * 128)()
if rank == 0:
    capi.check(lib.ek_hip_dist_unique_id(ident)); open(path + ".tmp", "wb").write(bytes(ident)); os.rename(path + ".tmp", path)
else:
    while not os.path.exists(path): time.sleep(0.05)
    ident.raw = open(path, "rb").read()
capi.check(lib.ek_hip_dist_init(rank, world, ident))
K = 1 << 20
g = capi.Buf.from_numpy(np.full(K, float(rank + 1), np.float32))
capi.check(lib.ek_hip_dist_all_reduce(g.ek, 0, ctypes.c_void_p(g.ptr), SZ(K)))
own = capi.Buf(np.float32, K // world)
src = capi.Buf.from_numpy(np.arange(K, dtype=np.float32) * (rank + 1))
capi.check(lib.ek_hip_dist_reduce_scatter(src.ek, 0, ctypes.c_void_p(own.ptr), ctypes.c_void_p(src.ptr), SZ(K // world)))
capi.sync()
tot = world * (world + 1) / 2
assert np.all(g.numpy() == tot)
c = K // world
assert np.array_equal(own.numpy(), np.arange(rank * c, (rank + 1) * c, dtype=np.float32) * tot)
capi.check(lib.ek_hip_dist_finalize())
print("ok", rank)
"""


def _gpu_count():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.gpu
@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs")
def test_two_gpus_through_the_c_abi(tmp_path):
    path = str(tmp_path / "rccl_id")
    procs = [subprocess.Popen([sys.executable, "-c", WORKER, ROOT, str(r), "2", path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0 and "ok" in so, se[-2000:]
