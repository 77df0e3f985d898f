# per-kernel breakdown of the headline step (HIP events per launch, eager repeat inside bench.py)
python bench.py --no-cpu-baseline --no-also --steps 20 --warmup 3 --profile-steps 5 "$@" | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['unit'], d['ms_per_step'], 'ms/step', d['config'].get('step_replay'))
for k in d['roofline']['kernels']: print('%-28s x%.0f  %.4f ms  %s TB/s' % (k['kernel'], k['launches_per_step'], k['avg_ms'], k['tb_s']))
"
