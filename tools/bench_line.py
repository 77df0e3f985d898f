"""one-screen summary of a bench.py JSON line (file name, or - for stdin)"""
import json, sys
src = sys.stdin.read() if sys.argv[1] == "-" else open(sys.argv[1]).read()
lines = [l for l in src.strip().splitlines() if l.startswith("{")]
if not lines:
    sys.exit("no JSON line")
d = json.loads(lines[-1])
print("HEADLINE", d["value"], d["unit"], d["ms_per_step"], "ms/step  n_gpus", d["n_gpus"], "scaling", d["scaling"], "pre_warm", d.get("pre_warm_s"), "parity", d.get("parity_checked"))
r = d.get("roofline")
if r:
    print("dominant", r["kernel"], "frac", r["frac"], "live", r.get("frac_live"), "rocprof", r.get("frac_rocprof"), r.get("rocprof_avg_us"), "traffic", r.get("traffic"), "whole", r["whole_step"]["frac"], r.get("trace_check"))
    for k in r["kernels"]:
        print("   %-34s x%.0f  %.4f ms  %s TB/s" % (k["kernel"], k["launches_per_step"], k["avg_ms"], k["tb_s"]))
for w, v in (d.get("also") or {}).items():
    print("  also %-26s %9.2f  %.4f ms  %s B/elt  dom %s" % (w, v["value"], v["ms_per_step"], v["bytes_per_elt"], v["dominant_kernel"]))
if d.get("weak"):
    print("  weak", d["weak"])
if d.get("cpu_baseline"):
    print("  cpu", d["cpu_baseline"].get("value"), d["cpu_baseline"].get("unit"), d["cpu_baseline"].get("cores"), d["cpu_baseline"].get("kind"))
