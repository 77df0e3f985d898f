# round 5, first GPU call: the suites, smoke, the driver's bench protocol, an A/B of the lock-free (compare-and-swap) early adjoint
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/pytest_1.log 2>&1; tail -15 gpurun_out/pytest_1.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
# the driver's protocol
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver.json 2> gpurun_out/bench_driver.err; tail -3 gpurun_out/bench_driver.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_driver.json").read().strip().splitlines()[-1])
print("HEADLINE", d["value"], d["unit"], d["ms_per_step"], "ms/step pre_warm", d.get("pre_warm_s"), "parity", d["parity_checked"])
r = d["roofline"]
print("dominant", r["kernel"], r["frac"], "whole", r["whole_step"]["frac"], r["whole_step"]["bytes_per_elt"], "B/elt", r.get("trace_check"))
for k in r["kernels"]: print("   %-32s x%.0f  %.4f ms  %s TB/s" % (k["kernel"], k["launches_per_step"], k["avg_ms"], k["tb_s"]))
for w, v in (d.get("also") or {}).items(): print("  also %-26s %9.2f  %.4f ms  %s B/elt  dom %s" % (w, v["value"], v["ms_per_step"], v["bytes_per_elt"], v["dominant_kernel"]))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"])
PY
# A/B: exchange locks (product) against compare-and-swap, one and two vectors per lane; libraries swapped in place, alternating
cp enoki_amd/libenoki-hip.so /tmp/base.so
for round in 1 2 3; do
  for v in base cas cas2; do
    if [ $v = base ]; then cp /tmp/base.so enoki_amd/libenoki-hip.so; else cp build/variants/libenoki-hip-$v.so enoki_amd/libenoki-hip.so; fi
    timeout 120 python tools/probe_early.py 26 20 $v 2>&1 | tail -1
  done
done | tee gpurun_out/probe_early_cas.txt
cp /tmp/base.so enoki_amd/libenoki-hip.so
# the 8 Mi-element shard step and the two-rank form through plain `python bench.py --gpus 2`
timeout 300 python bench.py --n 8388608 --steps 200 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('8Mi shard', d['value'], d['ms_per_step'])
for k in d['roofline']['kernels']: print('   %-32s x%.0f  %.4f ms' % (k['kernel'], k['launches_per_step'], k['avg_ms']))
"
