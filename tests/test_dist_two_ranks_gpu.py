"""The sharded bench path end to end on a device: two ranks (processes) share the one GPU of the test box, collectives go
through gloo (RCCL refuses two ranks on one device; on a multi-GPU node the same code runs with backend nccl = RCCL).
Exercises enoki_amd/dist.py on device buffers: index-range shards, the packed asynchronous all-reduce of y and the two
table gradients, max-over-ranks timing -- the all-reduced loss must equal the single-process loss up to summation order."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(workload, n, ranks, port, dump=None, extra=()):
    """`python bench.py --gpus <ranks>` exactly as the driver types it for one GPU -- NO launcher, no RANK in the environment: for
    ranks > 1 bench.py starts its own ranks (torch.distributed.run, rendezvous on 127.0.0.1) and, on a box with fewer GPUs than
    ranks, routes the collectives through gloo (the line says so).  `port` is unused since round 5 (bench.py picks a free one)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "3", "--warmup", "1", "--n", str(n),
           "--workload", workload, "--no-cpu-baseline", "--no-also", "--profile-steps", "1", "--pre-warm-s", "0.05"] + list(extra)
    if dump:
        cmd += ["--dump-gradients", dump]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 alone prints the JSON line"
    line = json.loads(lines[0])
    assert line["n_gpus"] == ranks and line["config"]["world_size"] == ranks, (line["n_gpus"], line["config"].get("world_size"))
    return line


@pytest.mark.parametrize("workload", ["cfg3b", "cfg3a"])
def test_two_ranks_match_one(workload, tmp_path):
    n = 1 << 22
    one = run_bench(workload, n, 1, 29611, dump=str(tmp_path / "one"))
    # (--scaling strong: the SAME n elements on two ranks, so that the single process is the yardstick)
    # cfg3b: ONE reduce-scatter for both gradient tables with the loss riding in an extra column; cfg3a: the loss only
    two = run_bench(workload, n, 2, 29612 if workload == "cfg3b" else 29613, dump=str(tmp_path / "two"),
                    extra=["--scaling", "strong"] + (["--reduce-scatter-grads"] if workload == "cfg3b" else []))
    if workload == "cfg3b":
        check_scattered_gradients(n, tmp_path)
        assert "reduce-scatter" in two["config"]["gradient_exchange"]
        # the record form (north_star: "grad accumulation finished by RCCL all-reduce"): ONE all-reduce, every rank holds all K bins
        full = run_bench(workload, n, 2, 0, dump=str(tmp_path / "full"), extra=["--scaling", "strong"])
        assert "all-reduce" in full["config"]["gradient_exchange"] and full["config"]["collectives_per_step"] == 1
        check_all_reduced_gradients(n, tmp_path)
    assert two["n_gpus"] == 2 and two["config"]["elements_per_gpu"] == n // 2
    assert two["config"]["collectives_per_step"] == 1
    assert "gloo" in two["config"]["backend"] or "nccl" in two["config"]["backend"]
    truth, bound = truth_y(workload, n)
    assert abs(one["result_y"] - truth) <= bound and abs(two["result_y"] - truth) <= bound, (one["result_y"], two["result_y"], truth, bound)
    assert two["value"] > 0 and two["scaling"] == "strong"


def test_default_is_the_contracts_strong_scaling_with_the_weak_result_alongside():
    """`python bench.py --gpus 2` without further options measures what BASELINE.md section 4 / north_star state -- --n elements IN
    TOTAL, n / 2 per rank, `"scaling": "strong"` -- and carries the weak measurement of the same run (every rank owns --n elements of
    an array of 2 x --n) as the labelled sub-record `weak`; `--scaling weak` makes that the `value` (round 5's default)"""
    n = 1 << 21
    two = run_bench("cfg3b", n, 2, 0)
    assert two["scaling"] == "strong" and two["config"]["elements_per_gpu"] == n // 2 and two["config"]["elements_total"] == n
    assert two["config"]["collectives_per_step"] == 1
    truth, bound = truth_y("cfg3b", n)
    assert abs(two["result_y"] - truth) <= bound, (two["result_y"], truth, bound)
    w = two["weak"]
    assert w["scaling"] == "weak" and w["elements_per_gpu"] == n and w["elements_total"] == 2 * n and w["value"] > 0
    one = run_bench("cfg3b", n, 1, 0)
    assert one["scaling"] == "strong" and one["config"]["elements_total"] == n and one["weak"] is None
    weak = run_bench("cfg3b", n, 2, 0, extra=["--scaling", "weak"])
    assert weak["scaling"] == "weak" and weak["config"]["elements_per_gpu"] == n and weak["config"]["elements_total"] == 2 * n
    truth2, bound2 = truth_y("cfg3b", 2 * n)
    assert abs(weak["result_y"] - truth2) <= bound2, (weak["result_y"], truth2, bound2)


def check_scattered_gradients(n, tmp_path):
    """the REDUCE-SCATTERED table gradients on the device: rank r holds bins [r K / 2, (r + 1) K / 2) of the sum over both shards;
    the owned slices, concatenated, are the single-process gradients -- both inside the per-bin class-D bound of the float64 sums"""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import cfg3b_truth, hash_u32, uniform_pm1
    K = 1 << 20
    A, B, x = uniform_pm1(K, 6), uniform_pm1(K, 7), uniform_pm1(n, 2)
    idx = (hash_u32(np.arange(n, dtype=np.uint64), 4) % np.uint32(K)).astype(np.uint32)
    t = cfg3b_truth(A, B, x, idx)
    single = np.load(tmp_path / "one" / "grad_rank0.npz")
    parts = sorted((np.load(tmp_path / "two" / f"grad_rank{r}.npz") for r in range(2)), key=lambda z: int(z["begin"]))
    assert int(parts[0]["begin"]) == 0 and int(parts[0]["end"]) == int(parts[1]["begin"]) and int(parts[1]["end"]) == K
    for g in ("gA", "gB"):
        whole = np.concatenate([p[g][: int(p["end"]) - int(p["begin"])] for p in parts])
        assert whole.shape == (K,)
        # two shards: every bin is the sum of two partial sums of at most cnt terms each
        assert np.all(np.abs(whole - t[g]) <= t[g + "_bound"] + 2.0 ** -24 * np.abs(t[g])), g
        assert np.all(np.abs(single[g] - t[g]) <= t[g + "_bound"]), g


def check_all_reduced_gradients(n, tmp_path):
    """default exchange: after ONE all-reduce every rank holds the full gradient tables -- identical on both ranks, inside the
    per-bin class-D bound of the float64 sums"""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import cfg3b_truth, hash_u32, uniform_pm1
    K = 1 << 20
    A, B, x = uniform_pm1(K, 6), uniform_pm1(K, 7), uniform_pm1(n, 2)
    idx = (hash_u32(np.arange(n, dtype=np.uint64), 4) % np.uint32(K)).astype(np.uint32)
    t = cfg3b_truth(A, B, x, idx)
    r0, r1 = (np.load(tmp_path / "full" / f"grad_rank{r}.npz") for r in range(2))
    for g in ("gA", "gB"):
        assert r0[g].shape == (K,) and np.array_equal(r0[g].view(np.uint32), r1[g].view(np.uint32)), g
        assert np.all(np.abs(r0[g] - t[g]) <= t[g + "_bound"] + 2.0 ** -24 * np.abs(t[g])), g


def truth_y(workload, n):
    """float64 value of the bench's loss on its synthetic inputs (seeds of SURVEY 8d) and 5 sigma of independent f32 roundings
    for summation orders of our depth (conftest.stat_sum_bound; two shards of half the depth are covered by it)"""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import hash_u32, hsum_depth, stat_sum_bound, uniform_pm1
    x = uniform_pm1(n, 2).astype(np.float64)
    if workload == "cfg3b":
        K = 1 << 20
        idx = (hash_u32(np.arange(n, dtype=np.uint64), 4) % np.uint32(K)).astype(np.int64)
        u = uniform_pm1(K, 6).astype(np.float64)[idx] * x + uniform_pm1(K, 7).astype(np.float64)[idx]
    else:
        u = uniform_pm1(n, 1).astype(np.float64) * x + uniform_pm1(n, 3).astype(np.float64)
    s = np.sin(u)
    return float(s.sum()), stat_sum_bound(s, hsum_depth(n) + 64)


def _gpu_count():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_two_gpus_rccl():
    """the first multi-GPU box exercises the real thing: one rank per GPU, backend nccl (= RCCL over xGMI), launched the
    way the driver launches it (torch.distributed.run)"""
    n = 1 << 24
    one = run_bench("cfg3b", n, 1, 29621)
    two = run_bench("cfg3b", n, 2, 0, extra=["--scaling", "strong"])
    assert "nccl" in two["config"]["backend"], two["config"]["backend"]
    assert two["n_gpus"] == 2 and two["config"]["collectives_per_step"] == 1
    truth, bound = truth_y("cfg3b", n)
    assert abs(one["result_y"] - truth) <= bound and abs(two["result_y"] - truth) <= bound
