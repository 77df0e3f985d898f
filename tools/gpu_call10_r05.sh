# round 5: the ring of pre-zeroed counter blocks (no 768-word fill in front of every partition) on / off; headline and 8 Mi shard,
# alternating, same box; then the suites that exercise the bucket path (ring on)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {   # label, n, env...
  label=$1; n=$2; shift 2
  env "$@" timeout 300 python bench.py --n $n --steps 300 --warmup 5 --no-cpu-baseline --no-also --pre-warm-s 0.4 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-12s n=%-9d %8.2f Gelem/s %.4f ms  ' % ('$label', $n, d['value'], d['ms_per_step']) + ' '.join('%s %.1f' % (k['kernel'].replace('bucket_', '')[:20], k['avg_ms'] * 1e3) for k in d['roofline']['kernels']))
"
}
for round in 1 2 3; do
  for n in 67108864 8388608; do
    run "ring" $n ENOKI_HIP_META_RING=1
    run "fill" $n ENOKI_HIP_META_RING=0
  done
done | tee gpurun_out/probe_meta_ring.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | grep "passed\|failed" | tail -3
