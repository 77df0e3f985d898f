# round 6: the suites, smoke, the driver's bench protocol, then ALL the rocprofv3 evidence (tools/profile_r06.sh) on the same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# (SKIP_SUITE=1: the suite has just run on this very tree in a call of its own)
[ "$SKIP_SUITE" = 1 ] || { timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/pytest_final.log 2>&1; tail -4 gpurun_out/pytest_final.log | head -3; }
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_r06.sh 2>&1 | tail -22
# the bench line LAST: it finds the stamped summaries of this very tree under profiles/ only after they are copied there by hand, so
# frac_rocprof / traffic of THIS line come from the previous copy when the kernels did not change; the summaries above are what gets committed
cp gpurun_out/rocprof_kernel_stats_r06.txt gpurun_out/rocprof_pmc_r06.txt profiles/ 2>/dev/null
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver.json 2> gpurun_out/bench_driver.err; tail -3 gpurun_out/bench_driver.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_driver.json").read().strip().splitlines()[-1])
print("HEADLINE", d["value"], d["unit"], d["ms_per_step"], "ms/step pre_warm", d.get("pre_warm_s"), "parity", d["parity_checked"])
r = d["roofline"]
print("dominant", r["kernel"], "frac", r["frac"], "rocprof", r["frac_rocprof"], r["rocprof_avg_us"], "traffic", r["traffic"], "whole", r["whole_step"]["frac"], r.get("trace_check"))
for k in r["kernels"]: print("   %-32s x%.0f  %.4f ms  %s TB/s" % (k["kernel"], k["launches_per_step"], k["avg_ms"], k["tb_s"]))
for w, v in (d.get("also") or {}).items(): print("  also %-26s %9.2f  %.4f ms  %s B/elt  dom %s" % (w, v["value"], v["ms_per_step"], v["bytes_per_elt"], v["dominant_kernel"]))
PY
timeout 300 python bench.py --n 8388608 --steps 300 --warmup 5 --no-cpu-baseline --no-also --pre-warm-s 0.3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('8Mi shard', d['value'], d['ms_per_step'], ' '.join('%s %.1f' % (k['kernel'][:22], k['avg_ms'] * 1e3) for k in d['roofline']['kernels']))
"
# round 6 extras: the second-wave chains, the forward + adjoint kernel under locks against fixed point (same box), the SQ counters of both
timeout 300 python tools/probe_second_wave.py 26 2>&1 | tee gpurun_out/probe_second_wave.txt
timeout 300 bash tools/ab_early.sh 2>&1 | tee gpurun_out/probe_early.txt
timeout 300 bash tools/profile_early.sh fixed PROBE_HINTS=3 > /dev/null 2>&1; timeout 300 bash tools/profile_early.sh locks PROBE_HINTS=1 > /dev/null 2>&1
