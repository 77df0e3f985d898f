# round 5: plain scatter_add(value, index) through the page partition (ENOKI_HIP_SCATTER_PAGED=1, new) against the count / scan /
# partition pipeline (=0); 64 Mi adds by table size, then cfg5 and the element-order cfg3b step; alternating; then the whole GPU suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for round in 1 2; do
  for p in 1 0; do
    echo "ENOKI_HIP_SCATTER_PAGED=$p"; ENOKI_HIP_SCATTER_PAGED=$p timeout 300 python tools/probe_scatter_sizes.py 2>&1 | grep "K=2^\(16\|18\|20\|22\)"
    for w in cfg5 cfg3b; do
      ENOKI_HIP_SCATTER_PAGED=$p ENOKI_HIP_BUCKET_ORDERED=$([ $w = cfg3b ] && echo 1 || echo 1) timeout 300 python bench.py --workload $w --steps 60 --warmup 3 --no-cpu-baseline --no-also --pre-warm-s 0.3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   %-6s %8.2f %s %.4f ms  ' % ('$w', d['value'], d['unit'], d['ms_per_step']) + ' '.join('%s %.1f' % (k['kernel'][:24], k['avg_ms'] * 1e3) for k in d['roofline']['kernels'][:6]))
"
    done
  done
done | tee gpurun_out/probe_scatter_paged.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | grep "passed\|failed\|^FAILED\|^E  " | tail -12
