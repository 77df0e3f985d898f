# round 5: cfg3a after the chain learned a second output (ek_hip_map_chain_product) and reductions read through an arithmetic node that
# only a sibling map holds: the tests of the new paths, the device fuzzers, then cfg3a / cfg2 / cfg3b bench lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_chain_gpu.py tests/test_deferred_map_gpu.py tests/test_headline_parity_gpu.py tests/test_reference_autodiff_gpu.py tests/test_python_api_gpu.py -m gpu -q --timeout 600 2>&1 | grep "passed\|failed\|^FAILED\|^E  " | tail -15
for w in cfg3a cfg2 cfg3b; do
  timeout 300 python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline --no-also --pre-warm-s 0.3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   %-6s %8.2f %s %.4f ms  B/elt %s  ' % ('$w', d['value'], d['unit'], d['ms_per_step'], d['roofline']['whole_step']['bytes_per_elt']) + ' '.join('%s %.1f' % (k['kernel'][:24], k['avg_ms'] * 1e3) for k in d['roofline']['kernels'][:6]))
"
done | tee gpurun_out/probe_cfg3a.txt
