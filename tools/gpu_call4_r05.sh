# round 5, fourth GPU call: A/B of (a) the per-bucket fold inside the forward kernel, (b) interleaved tiles in the page partition,
# each on the headline (64 Mi) and on the 8 Mi-element shard; alternating, same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {   # label, n, env...
  label=$1; n=$2; shift 2
  env "$@" timeout 300 python bench.py --n $n --steps 300 --warmup 5 --no-cpu-baseline --no-also --pre-warm-s 0.4 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-34s n=%-9d %8.2f Gelem/s %.4f ms  ' % ('$label', $n, d['value'], d['ms_per_step']) + ' '.join('%s %.1f' % (k['kernel'].replace('bucket_', '')[:20], k['avg_ms'] * 1e3) for k in d['roofline']['kernels']), 'parity', d['parity_checked'])
"
}
for round in 1 2; do
  for n in 67108864 8388608; do
    run "fold=kernel interleave=0" $n ENOKI_HIP_FOLD_IN_KERNEL=1 ENOKI_HIP_PAGE_INTERLEAVE=0
    run "fold=launch interleave=0" $n ENOKI_HIP_FOLD_IN_KERNEL=0 ENOKI_HIP_PAGE_INTERLEAVE=0
    run "fold=launch interleave=1" $n ENOKI_HIP_FOLD_IN_KERNEL=0 ENOKI_HIP_PAGE_INTERLEAVE=1
    run "fold=kernel interleave=1" $n ENOKI_HIP_FOLD_IN_KERNEL=1 ENOKI_HIP_PAGE_INTERLEAVE=1
  done
done | tee gpurun_out/probe_fold_interleave.txt
ENOKI_HIP_PAGE_INTERLEAVE=1 timeout 900 python -m pytest tests/test_bucketed_gpu.py tests/test_bucket_ordered_gpu.py tests/test_neighbours_gpu.py -q --timeout 600 2>&1 | tail -5
timeout 900 python -m pytest tests/test_bucketed_gpu.py tests/test_bucket_ordered_gpu.py tests/test_neighbours_gpu.py tests/test_deferred_map_gpu.py tests/test_dist_two_ranks_gpu.py tests/test_dist_capi_gpu.py -q --timeout 600 2>&1 | tail -5
