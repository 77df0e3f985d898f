# round 5: vectors per lane / store kind of the chain kernel with a tail (cfg3a backward pass), same-call A/B by swapping the library;
# then what the operator spelling of cfg3a launches
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp enoki_amd/libenoki-hip.so /tmp/base.so
run() {
  timeout 300 python bench.py --workload cfg3a --steps 40 --warmup 5 --no-cpu-baseline --no-also --pre-warm-s 0.3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   %-6s %8.2f %s %.4f ms  B/elt %s  ' % ('$1', d['value'], d['unit'], d['ms_per_step'], d['roofline']['whole_step']['bytes_per_elt']) + ' '.join('%s %.1f' % (k['kernel'][:24], k['avg_ms'] * 1e3) for k in d['roofline']['kernels'][:6]))
"
}
for round in 1 2; do
  for v in base u2 u4 u1t u2t; do
    if [ $v = base ]; then cp /tmp/base.so enoki_amd/libenoki-hip.so; else cp build/variants/libenoki-hip-$v.so enoki_amd/libenoki-hip.so; fi
    run $v
  done
done | tee gpurun_out/probe_chain_tail.txt
cp /tmp/base.so enoki_amd/libenoki-hip.so
timeout 300 python - <<'PY' | tee -a gpurun_out/probe_chain_tail.txt
import json, numpy as np
import enoki_amd.hip as ek, enoki_amd.hip_autodiff as ad
ek.hip_init(0)
n = 1 << 22
rng = np.random.default_rng(1)
a, x, b = (rng.uniform(-1, 1, n).astype(np.float32) for _ in range(3))
xd = ad.Float32(x)
def step(spell):
    da, db = ad.Float32(a), ad.Float32(b)
    ad.set_requires_gradient(da); ad.set_requires_gradient(db)
    y = ad.hsum(ad.sin(da * xd + db)) if spell else ad.hsum(ad.sin(ad.fmadd(da, xd, db)))
    ad.backward(y)
    return ad.gradient(da).numpy(), ad.gradient(db).numpy()
for spell in (0, 1):
    step(spell)
    ek.hip_profile_begin(); ga, gb = step(spell); prof = json.loads(ek.hip_profile_end())
    print("operators" if spell else "fmadd", {k["kernel"]: k["launches"] for k in prof if k["launches"]})
    u = (a * x + b) if spell else None
PY
