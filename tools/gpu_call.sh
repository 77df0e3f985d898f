# ONE parametrised GPU call (replaces the per-experiment tools/gpu_call*_r05.sh of round 5):
#   gpurun --timeout T -- 'bash tools/gpu_call.sh <step> [<step> ...]'
# steps:  probe:<tools/probe_x.py>[:args]   suite[:<pytest args>]   smoke   bench[:<bench args>]   profile[:quick]   sh:<command>
# every step's output is tee'd into gpurun_out/<step name>.log; the call never stops at a failing step.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for step in "$@"; do
  kind=${step%%:*}; arg=""; [ "$kind" != "$step" ] && arg=${step#*:}
  echo "=== $step"
  case $kind in
    probe)   f=${arg%%:*}; a=""; [ "$f" != "$arg" ] && a=${arg#*:}
             timeout 600 python -u tools/$f $a 2>&1 | tee gpurun_out/$(basename $f .py).txt | tail -60 ;;
    suite)   timeout 1800 python -m pytest ${arg:-tests} -m gpu -q --timeout 900 > gpurun_out/pytest.log 2>&1; grep "passed\|failed\|^FAILED\|^E  " gpurun_out/pytest.log | tail -15 ;;
    smoke)   timeout 180 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ;;
    bench)   timeout 900 python bench.py $arg > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -2 gpurun_out/bench.err; python tools/bench_line.py gpurun_out/bench.json ;;
    profile) bash tools/profile_r06.sh $arg 2>&1 | tail -30 ;;
    sh)      bash -c "$arg" 2>&1 | tail -60 ;;
    *)       echo "unknown step $step" ;;
  esac
done
