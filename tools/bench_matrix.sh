# every bench.py entry point once, small and quick: catches flag combinations that break (GPU box: bash tools/bench_matrix.sh)
run() { echo "== $*"; timeout 300 python bench.py --no-cpu-baseline --no-also --steps 3 --warmup 1 --profile-steps 1 "$@" 2>&1 | grep "^{\"metric\|Error\|error" | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   ', d['value'], d['unit'], d['ms_per_step'], 'ms', d['config'].get('step_replay', ''), 'n_gpus', d['n_gpus'])
    else: print('   ', l.strip()[:200])"; }
for w in cfg3b cfg3a cfg2 cfg4 cfg4_unfused cfg5; do run --workload $w; done
run --eager
run --deterministic
run --n 1000003
run --n 4096
run --workload cfg3a --n 1000003 --eager
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 run --n 4194304
