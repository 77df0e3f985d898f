# round 5: where the bucket-ordered path starts to pay -- the cfg3b step by input size, bucket order on / off (K = 1 Mi table entries)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for n in 262144 524288 1048576 2097152 4194304 8388608 16777216; do
  for bo in 1 0; do
    ENOKI_HIP_BUCKET_ORDERED=$bo timeout 200 python bench.py --n $n --steps 100 --warmup 10 --no-cpu-baseline --no-also --pre-warm-s 0.1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
ks = d['roofline']['kernels']
print('n=%9d bucket_ordered=$bo  %8.2f Gelem/s  step %.4f ms  kernels %.1f us  ' % ($n, d['value'], d['ms_per_step'], sum(k['avg_ms'] * k['launches_per_step'] for k in ks) * 1e3) + ' '.join('%s %.1f' % (k['kernel'][:20], k['avg_ms'] * 1e3) for k in ks[:5]))
"
  done
done | tee gpurun_out/probe_sizes.txt
