# round 5: concurrent slices of a split table (K = 16 Mi, 8 Mi) on / off, alternating, same box; then the sliced-table tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {   # label, workload, env...
  label=$1; w=$2; shift 2
  env "$@" timeout 300 python bench.py --workload $w --steps 100 --warmup 3 --no-cpu-baseline --no-also --pre-warm-s 0.3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-22s %-14s %8.2f Gelem/s %.4f ms' % ('$label', '$w', d['value'], d['ms_per_step']))
"
}
for round in 1 2; do
  for w in cfg3b_K16Mi cfg3b_K4Mi; do
    run "slices concurrent" $w ENOKI_HIP_CONCURRENT_SLICES=1
    run "slices back to back" $w ENOKI_HIP_CONCURRENT_SLICES=0
  done
done | tee gpurun_out/probe_slices.txt
timeout 900 python -m pytest tests/test_bucketed_gpu.py tests/test_bucket_ordered_gpu.py tests/test_neighbours_gpu.py tests/test_headline_parity_gpu.py -q --timeout 600 -k "slice or K4Mi or large or big or table" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_bucketed_gpu.py tests/test_bucket_ordered_gpu.py -q --timeout 600 2>&1 | tail -3
