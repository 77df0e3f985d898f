"""Summarise rocprofv3 sqlite outputs (gpurun_out/prof_*/**/*.db) into the text files kept under profiles/.

    python tools/rocprof_summary.py kernels gpurun_out/prof_kt   > profiles/rocprof_kernel_stats_rNN.txt
    python tools/rocprof_summary.py pmc gpurun_out/prof_fetch gpurun_out/prof_write > profiles/rocprof_pmc_rNN.txt

PMC correction (MI355X_MICROARCH.md, section HBM): on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide
coalesced streaming read, so the read side is doubled before comparing with byte counts; WRITE_SIZE is reported
as is (uncalibrated).  Both counters are in KiB."""
import glob
import re
import os
import sqlite3
import sys


def dbs(d):
    return glob.glob(d + "/**/*.db", recursive=True)


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("ek::", "")[:110]


def kernels(d):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from enoki_amd import _build
    print(f"# kernels_sha16: {_build.kernels_sha16()}")
    for f in dbs(d):
        cur = sqlite3.connect(f).cursor()
        print(f"# rocprofv3 --kernel-trace --stats  ({f})")
        print(f"{'kernel':112s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'%':>7s}")
        rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
        for name, calls, total, avg, pct in rows:
            print(f"{short(name):112s} {calls:6d} {total:12.1f} {avg:10.2f} {pct:7.2f}")
        # one step of the profiled workload = one launch of the partition kernel (bucket-ordered workloads): every kernel that is
        # launched at least once per step, summed over ALL its launches and divided by the number of steps -- what the kernels of one
        # step take back to back on an in-order stream (bench.py compares its ms_per_step with this: trace_check)
        steps = max((calls for name, calls, *_ in rows if "k_page_partition" in name), default=0)
        if steps:
            per_step = sum(total for name, calls, total, *_ in rows if calls >= steps) / steps
            print(f"# steps: {steps}")
            print(f"# step_sum_us: {per_step:.2f}")


def pmc(dirs):
    agg = {}
    for d in dirs:
        for f in dbs(d):
            cur = sqlite3.connect(f).cursor()
            for name, counter, value, grid in cur.execute("select kernel_name, counter_name, value, grid_size from counters_collection"):
                if grid < 65536:
                    continue
                a = agg.setdefault((short(name), grid), {})
                s = a.setdefault(counter, [0, 0.0])
                s[0] += 1; s[1] += value
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from enoki_amd import _build
    print(f"# kernels_sha16: {_build.kernels_sha16()}")
    print("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); per launch averages; large grids only")
    print("# hbm_read = 2 * FETCH_SIZE KiB (gfx950 correction), hbm_write = WRITE_SIZE KiB")
    print(f"{'kernel':112s} {'grid':>10s} {'launches':>8s} {'read_MB':>10s} {'write_MB':>10s}")
    for (name, grid), c in sorted(agg.items(), key=lambda kv: -sum(v[1] for v in kv[1].values())):
        fe = c.get("FETCH_SIZE", [0, 0.0]); wr = c.get("WRITE_SIZE", [0, 0.0])
        rd = 2 * fe[1] / max(fe[0], 1) * 1024 / 1e6
        wm = wr[1] / max(wr[0], 1) * 1024 / 1e6
        print(f"{name:112s} {grid:10d} {max(fe[0], wr[0]):8d} {rd:10.1f} {wm:10.1f}")


def raw(dirs):
    """per-launch averages of whatever counters the passes collected (large grids only)"""
    agg = {}
    for d in dirs:
        for f in dbs(d):
            cur = sqlite3.connect(f).cursor()
            for name, counter, value, grid in cur.execute("select kernel_name, counter_name, value, grid_size from counters_collection"):
                if grid < 65536:
                    continue
                s = agg.setdefault(short(name), {}).setdefault(counter, [0, 0.0])
                s[0] += 1; s[1] += value
    for name, c in agg.items():
        print(name)
        for counter, (cnt, total) in sorted(c.items()):
            print(f"    {counter:28s} launches {cnt:5d}  avg {total / max(cnt, 1):16.1f}")


if __name__ == "__main__":
    if sys.argv[1] == "kernels":
        kernels(sys.argv[2])
    elif sys.argv[1] == "raw":
        raw(sys.argv[2:])
    else:
        pmc(sys.argv[2:])
