# round 5: the partition workgroups' own page directory in the LDS (ENOKI_HIP_WDIR_LDS=1, new) against global memory + the wait for all of
# the workgroup's stores (=0); alternating; cfg3b at 64 Mi and 8 Mi, masked, cfg5; then the suites of the bucket paths
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {
  ENOKI_HIP_WDIR_LDS=$1 timeout 300 python bench.py --workload $2 --n $3 --steps $4 --warmup 5 --no-cpu-baseline --no-also --pre-warm-s 0.5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   wdir_lds=$1 %-6s n=%9d %8.2f %s %.4f ms  ' % ('$2', $3, d['value'], d['unit'], d['ms_per_step']) + ' '.join('%s %.1f' % (k['kernel'][:24], k['avg_ms'] * 1e3) for k in d['roofline']['kernels'][:5]))
"
}
for round in 1 2 3; do
  for b in 1 0; do run $b cfg3b 67108864 60; done
done | tee gpurun_out/probe_wdir_lds.txt
for round in 1 2; do for b in 1 0; do run $b cfg3b 8388608 300; done; done | tee -a gpurun_out/probe_wdir_lds.txt
for b in 1 0; do run $b cfg3b_masked 67108864 60; run $b cfg5 67108864 40; done | tee -a gpurun_out/probe_wdir_lds.txt
timeout 900 python -m pytest tests/test_bucketed_gpu.py tests/test_neighbours_gpu.py tests/test_headline_parity_gpu.py tests/test_bucket_ordered_gpu.py tests/test_cfg5_gpu.py -m gpu -q --timeout 600 2>&1 | grep "passed\|failed\|^FAILED\|^E  " | tail -12
