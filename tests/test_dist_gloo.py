"""The N>1 path on CPU: two processes, gloo backend, world size 2 (no GPU).

What is exercised is the product's sharding layer (enoki_amd/dist.py): index-range partition, the packed
single all-reduce (scalar hsum + K-element table gradients), max-over-ranks timing.  The per-shard compute is
done with the CPU oracle (tests may use it), and the all-reduced result must equal the unsharded oracle result
up to the order-dependent summation bound."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["EK_ROOT"]); sys.path.insert(0, os.path.join(os.environ["EK_ROOT"], "tests"))
import torch, torch.distributed as dist
from enoki_amd import dist as ekd
import oracle_lib
from conftest import hash_u32, uniform_pm1

rank, local_rank, world = ekd.init("gloo")
assert world == 2 and dist.get_backend() == "gloo"
N, K = 200003, 4096
begin, end = ekd.shard_range(N, rank, world)
assert ekd.shard_range(N, 0, world)[1] == ekd.shard_range(N, 1, world)[0] and ekd.shard_range(N, world - 1, world)[1] == N
idx_all = np.arange(N, dtype=np.uint64)
x = uniform_pm1(N, 2)[begin:end]
idx = (hash_u32(idx_all, 4) % np.uint32(K)).astype(np.uint32)[begin:end]
A, B = uniform_pm1(K, 6), uniform_pm1(K, 7)          # replicated tables
P = oracle_lib.port()
y, gA, gB, _ = P.cfg3b(A, B, np.ascontiguousarray(x), np.ascontiguousarray(idx))   # local shard

packer = ekd.Packer([1, K, K], "cpu")
packer.pack([torch.tensor([y], dtype=torch.float32), torch.from_numpy(gA), torch.from_numpy(gB)])
ty, tgA, tgB = packer.all_reduce()                     # ONE collective, asynchronous
packer.wait_all()
t = ekd.max_over_ranks(float(rank + 1))
ekd.barrier()
if rank == 0:
    np.savez(os.environ["EK_OUT"], y=ty.numpy(), gA=tgA.numpy(), gB=tgB.numpy(), tmax=np.array([t]))
dist.destroy_process_group()
'''


def test_sharded_cfg3b_allreduce_matches_unsharded(tmp_path):
    out = tmp_path / "out.npz"
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, EK_ROOT=ROOT, EK_OUT=str(out), MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2"))
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    z = np.load(out)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from conftest import hash_u32, uniform_pm1
    N, K = 200003, 4096
    x = uniform_pm1(N, 2); idx = (hash_u32(np.arange(N, dtype=np.uint64), 4) % np.uint32(K)).astype(np.uint32)
    A, B = uniform_pm1(K, 6), uniform_pm1(K, 7)
    from conftest import cfg3b_truth
    y, gA, gB, _ = oracle_lib.port().cfg3b(A, B, x, idx)
    t = cfg3b_truth(A, B, x, idx)
    # class D against float64: the shards are summed by the CPU checker (8 lane-wise accumulators each), then added
    assert abs(float(z["y"][0]) - t["y"]) <= t["y_bound_reference"] and abs(float(z["y"][0]) - y) <= 2 * t["y_bound_reference"]
    for name, got, whole in (("gA", z["gA"], gA), ("gB", z["gB"], gB)):
        assert np.all(np.abs(got - t[name]) <= t[name + "_bound"]), name           # cnt * sum|terms| * 2^-24 per bin
        assert np.all(np.abs(got - whole) <= 2 * t[name + "_bound"]), name
    assert z["tmax"][0] == 2.0


def test_shard_range_partitions_exactly():
    from enoki_amd.dist import shard_range
    for n in (0, 1, 7, 64, 1 << 26, (1 << 26) + 5):
        for world in (1, 2, 3, 4, 8):
            edges = [shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in edges]
            assert max(sizes) - min(sizes) <= 1


WORKER_SHARDED = r'''
import os, sys, ctypes
import numpy as np
sys.path.insert(0, os.environ["EK_ROOT"]); sys.path.insert(0, os.path.join(os.environ["EK_ROOT"], "tests"))
import torch, torch.distributed as dist
from enoki_amd import dist as ekd
import oracle_lib
from conftest import hash_u32, uniform_pm1

P = oracle_lib.port()


class HostArrays:
    """the array module the sharding layer drives: here the CPU oracle on numpy arrays (the device modules have the same names)"""
    hsum = staticmethod(lambda x: np.array([P.reduce("hsum", x)], x.dtype))
    hprod = staticmethod(lambda x: np.array([P.reduce("hprod", x)], x.dtype))
    hmax = staticmethod(lambda x: np.array([P.reduce("hmax", x)], x.dtype))
    hmin = staticmethod(lambda x: np.array([P.reduce("hmin", x)], x.dtype))
    count = staticmethod(lambda m: int(np.count_nonzero(m)))
    detach = staticmethod(lambda x: x)
    gradient = staticmethod(lambda t: t.grad)


class Table:
    def __init__(self, values): self.values, self.grad = values, None


rank, local_rank, world = ekd.init("gloo")
N, K = 200003, 4096
sh = ekd.Sharded(HostArrays, N, device="cpu")
assert (sh.begin, sh.end) == ekd.shard_range(N, rank, world)

# ---- cfg3b: loss + two table gradients -> ONE float32 all-reduce, no explicit packing ----
x = uniform_pm1(N, 2)[sh.begin:sh.end]
idx = (hash_u32(np.arange(N, dtype=np.uint64), 4) % np.uint32(K)).astype(np.uint32)[sh.begin:sh.end]
A, B = Table(uniform_pm1(K, 6)), Table(uniform_pm1(K, 7))
u = P.ternary("fmadd", A.values[idx], np.ascontiguousarray(x), B.values[idx])
_, A.grad, B.grad, _ = P.cfg3b(A.values, B.values, np.ascontiguousarray(x), np.ascontiguousarray(idx))
y = sh.hsum(P.unary("sin", u)); gA = sh.gradient(A); gB = sh.gradient(B)
before = sh.exchange.collectives
plan = sh.flush()
assert sh.exchange.collectives - before == 1, "loss and gradients must share one collective"
res = {"y": y.tensor().numpy().copy(), "gA": gA.tensor().numpy().copy(), "gB": gB.tensor().numpy().copy()}
plan.run()                                                  # replay on the same sources (step-graph path)
assert np.array_equal(y.tensor().numpy(), res["y"]) and np.array_equal(gA.tensor().numpy(), res["gA"])

# ---- cfg4: shard-local permutation, count / max / min / any / all -> one int64 and two float32 collectives ----
res4 = 64
n4 = res4 * res4
lin = np.linspace(-1.2, 1.2, res4, dtype=np.float32)
gx, gy = np.tile(lin, res4), np.repeat(lin, res4)
b4, e4 = ekd.shard_range(n4, rank, world)
rng = np.random.default_rng(100 + rank)
perm = (rng.permutation(e4 - b4)).astype(np.uint32)         # shard-local indices
mask = np.ones(e4 - b4, np.uint8)
img = np.full(e4 - b4, -1.0, np.float32); hc = ctypes.c_uint64()
p_ = lambda a: a.ctypes.data_as(ctypes.c_void_p)
assert P.lib.orc_cfg4(p_(np.ascontiguousarray(gx[b4:e4])), p_(np.ascontiguousarray(gy[b4:e4])), p_(perm), p_(mask), ctypes.c_size_t(e4 - b4), p_(img),
                      ctypes.byref(hc)) == 0
sh4 = ekd.Sharded(HostArrays, n4, device="cpu")
hits = sh4.count(img >= 0); brightest = sh4.hmax(img); darkest = sh4.hmin(img)
anyhit = sh4.any(img >= 0); allhit = sh4.all(img >= 0)
before = sh4.exchange.collectives
sh4.flush()
assert sh4.exchange.collectives - before == 3             # int64 sum, float32 max, float32 min
res.update(hits=np.array([hits.item()]), hmax=brightest.tensor().numpy().copy(), hmin=darkest.tensor().numpy().copy(),
           anyhit=np.array([anyhit.item()]), allhit=np.array([allhit.item()]), local_hits=np.array([hc.value]))
ekd.barrier()
if rank == 0:
    np.savez(os.environ["EK_OUT"], **res)
dist.destroy_process_group()
'''


def test_library_level_sharding_cfg3b_and_cfg4(tmp_path):
    """enoki_amd.dist.Sharded: hsum / hmax / hmin / count / any / all / gradient(table) register themselves with the step's
    Exchange; flush() = one all-reduce per (dtype, reduction); no explicit Packer anywhere"""
    out = tmp_path / "out.npz"
    script = tmp_path / "worker.py"
    script.write_text(WORKER_SHARDED)
    env = dict(os.environ, EK_ROOT=ROOT, EK_OUT=str(out), MASTER_ADDR="127.0.0.1", MASTER_PORT="29543")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2"))
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    z = np.load(out)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes
    import oracle_lib
    from conftest import cfg3b_truth, hash_u32, uniform_pm1
    N, K = 200003, 4096
    x = uniform_pm1(N, 2); idx = (hash_u32(np.arange(N, dtype=np.uint64), 4) % np.uint32(K)).astype(np.uint32)
    t = cfg3b_truth(uniform_pm1(K, 6), uniform_pm1(K, 7), x, idx)
    assert abs(float(z["y"][0]) - t["y"]) <= t["y_bound_reference"]
    assert np.all(np.abs(z["gA"] - t["gA"]) <= t["gA_bound"]) and np.all(np.abs(z["gB"] - t["gB"]) <= t["gB_bound"])
    # cfg4: the unsharded image has the same multiset of shaded values (the permutation only moves pixels inside a shard)
    P = oracle_lib.port()
    res4 = 64; n4 = res4 * res4
    lin = np.linspace(-1.2, 1.2, res4, dtype=np.float32)
    gx, gy = np.tile(lin, res4), np.repeat(lin, res4)
    img = np.full(n4, -1.0, np.float32); hc = ctypes.c_uint64()
    p_ = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    perm = np.arange(n4, dtype=np.uint32); mask = np.ones(n4, np.uint8)
    assert P.lib.orc_cfg4(p_(gx), p_(gy), p_(perm), p_(mask), ctypes.c_size_t(n4), p_(img), ctypes.byref(hc)) == 0
    assert int(z["hits"][0]) == hc.value and z["hmax"][0] == img.max() and z["hmin"][0] == img.min()
    assert bool(z["anyhit"][0]) and not bool(z["allhit"][0])


WORKER_SCATTERED = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["EK_ROOT"]); sys.path.insert(0, os.path.join(os.environ["EK_ROOT"], "tests"))
import torch, torch.distributed as dist
from enoki_amd import dist as ekd
import oracle_lib
from conftest import hash_u32, uniform_pm1

P = oracle_lib.port()


class HostArrays:
    hsum = staticmethod(lambda x: np.array([P.reduce("hsum", x)], x.dtype))
    detach = staticmethod(lambda x: x)
    gradient = staticmethod(lambda t: t.grad)


class Table:
    def __init__(self, values): self.values, self.grad = values, None


rank, local_rank, world = ekd.init("gloo")
N, K = 200003, int(os.environ["EK_K"])                     # K = 4099: not a multiple of the world size (ragged last slice)
sh = ekd.Sharded(HostArrays, N, device="cpu")
x = np.ascontiguousarray(uniform_pm1(N, 2)[sh.begin:sh.end])
idx = np.ascontiguousarray((hash_u32(np.arange(N, dtype=np.uint64), 4) % np.uint32(K)).astype(np.uint32)[sh.begin:sh.end])
A, B = Table(uniform_pm1(K, 6)), Table(uniform_pm1(K, 7))
yl, A.grad, B.grad, _ = P.cfg3b(A.values, B.values, x, idx)
y = sh.reduce(np.array([yl], np.float32))
gA, gB = sh.gradient(A, scattered=True), sh.gradient(B, scattered=True)
before = sh.exchange.collectives
plan = sh.flush()
# ONE collective: both tables on one reduce-scatter, the loss in an extra column of it
assert sh.exchange.collectives - before == 1, sh.exchange.collectives - before
c = -(-K // world)
assert gA.owned == (min(rank * c, K), min((rank + 1) * c, K)) and gA.tensor().numel() == gA.owned[1] - gA.owned[0]
first = (gA.tensor().numpy().copy(), gB.tensor().numpy().copy())
plan.run()                                                 # replay on the same sources (step-graph path)
assert np.array_equal(gA.tensor().numpy(), first[0]) and np.array_equal(gB.tensor().numpy(), first[1])
fullA = sh.gather_scattered(gA, K).numpy()
fullB = sh.gather_scattered(gB, K).numpy()
assert np.array_equal(fullA[gA.owned[0]:gA.owned[1]], first[0])
ekd.barrier()
np.savez(os.environ["EK_OUT"] + f".{rank}.npz", y=y.tensor().numpy(), gA=first[0], gB=first[1], lo=np.array([gA.owned[0]]),
         fullA=fullA, fullB=fullB)
dist.destroy_process_group()
'''


import pytest  # noqa: E402


@pytest.mark.parametrize("world,K", [(2, 4096), (4, 4096), (4, 4099), (3, 4099)])
def test_reduce_scattered_table_gradients_equal_the_unsharded_oracle(tmp_path, world, K):
    """gradient(table, scattered=True): rank r ends up with bins [r c, (r + 1) c) of the global gradient (c = ceil(K / P)),
    one reduce-scatter for both tables; the slices of all ranks concatenate to the unsharded result, and the all-gather of
    gather_scattered() reproduces it on every rank"""
    out = tmp_path / "out"
    script = tmp_path / "worker.py"
    script.write_text(WORKER_SCATTERED)
    env = dict(os.environ, EK_ROOT=ROOT, EK_OUT=str(out), EK_K=str(K), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29550 + world + K % 7))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world)))
             for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import cfg3b_truth, hash_u32, uniform_pm1
    N = 200003
    x = uniform_pm1(N, 2); idx = (hash_u32(np.arange(N, dtype=np.uint64), 4) % np.uint32(K)).astype(np.uint32)
    t = cfg3b_truth(uniform_pm1(K, 6), uniform_pm1(K, 7), x, idx)
    gA, gB = np.zeros(K, np.float32), np.zeros(K, np.float32)
    covered = 0
    for r in range(world):
        z = np.load(str(out) + f".{r}.npz")
        lo = int(z["lo"][0])
        gA[lo:lo + z["gA"].size] = z["gA"]; gB[lo:lo + z["gB"].size] = z["gB"]
        covered += z["gA"].size
        assert abs(float(z["y"][0]) - t["y"]) <= t["y_bound_reference"]
        assert np.all(np.abs(z["fullA"] - t["gA"]) <= t["gA_bound"]) and np.all(np.abs(z["fullB"] - t["gB"]) <= t["gB_bound"])
    assert covered == K
    assert np.all(np.abs(gA - t["gA"]) <= t["gA_bound"]) and np.all(np.abs(gB - t["gB"]) <= t["gB_bound"])
