# round 5: class weights at gain 1/8 -- what they settle at on this box, cfg3b with and without (alternating), then the whole GPU suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/probe_xcd_balance.py 2>&1 | tee gpurun_out/probe_xcd_state.txt | tail -6
run() {
  ENOKI_HIP_XCD_BALANCE=$1 timeout 300 python bench.py --workload $2 --n $3 --steps 60 --warmup 5 --no-cpu-baseline --no-also --pre-warm-s 0.5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   balance=$1 %-6s n=%9d %8.2f %s %.4f ms  ' % ('$2', $3, d['value'], d['unit'], d['ms_per_step']) + ' '.join('%s %.1f' % (k['kernel'][:24], k['avg_ms'] * 1e3) for k in d['roofline']['kernels'][:5]))
"
}
for round in 1 2 3; do
  for b in 1 0; do run $b cfg3b 67108864; done
done | tee gpurun_out/probe_xcd_balance.txt
timeout 1400 python -m pytest tests -m gpu -q --timeout 900 2>&1 | grep "passed\|failed\|^FAILED\|^E  " | tail -12
