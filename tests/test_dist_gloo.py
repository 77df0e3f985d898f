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
    y, gA, gB, _ = oracle_lib.port().cfg3b(A, B, x, idx)
    assert abs(float(z["y"][0]) - y) <= N * 2.0 ** -23 * N
    cnt = np.bincount(idx, minlength=K) + 1
    assert np.all(np.abs(z["gA"] - gA) <= cnt * cnt * 2.0 ** -24)
    assert np.all(np.abs(z["gB"] - gB) <= cnt * cnt * 2.0 ** -24)
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
