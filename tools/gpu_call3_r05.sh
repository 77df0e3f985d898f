# round 5, third GPU call: suites, the driver's bench protocol, page size for small inputs, the rocprofv3 evidence of the headline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/pytest_3.log 2>&1; tail -25 gpurun_out/pytest_3.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver.json 2> gpurun_out/bench_driver.err; tail -3 gpurun_out/bench_driver.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_driver.json").read().strip().splitlines()[-1])
print("HEADLINE", d["value"], d["unit"], d["ms_per_step"], "ms/step pre_warm", d.get("pre_warm_s"), "parity", d["parity_checked"])
r = d["roofline"]
print("dominant", r["kernel"], r["frac"], "whole", r["whole_step"]["frac"], r["whole_step"]["bytes_per_elt"], "B/elt", r.get("trace_check"))
for k in r["kernels"]: print("   %-32s x%.0f  %.4f ms  %s TB/s" % (k["kernel"], k["launches_per_step"], k["avg_ms"], k["tb_s"]))
for w, v in (d.get("also") or {}).items(): print("  also %-26s %9.2f  %.4f ms  %s B/elt  dom %s" % (w, v["value"], v["ms_per_step"], v["bytes_per_elt"], v["dominant_kernel"]))
PY
for ps in 6 5 6 5; do
  for n in 8388608 16777216; do
    ENOKI_HIP_PAGE_SHIFT=$ps timeout 300 python bench.py --n $n --steps 300 --warmup 5 --no-cpu-baseline --no-also --pre-warm-s 0.3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('page_shift $ps n $n', d['value'], 'Gelem/s', d['ms_per_step'], 'ms', ' '.join('%s %.1f' % (k['kernel'][:18], k['avg_ms'] * 1e3) for k in d['roofline']['kernels']))
"
  done
done | tee gpurun_out/probe_page_shift.txt
bash tools/profile_r05.sh quick 2>&1 | tail -25
