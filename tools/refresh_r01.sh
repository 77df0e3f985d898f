# Refreshes everything under profiles/ that depends on the cfg3b step (run on the GPU box: bash tools/refresh_r01.sh)
R=$GRAFT_REPO_ROOT
bash $R/tools/profile_r01.sh > gpurun_out/profile.log 2>&1
cd $R
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
for w in cfg3a cfg2 cfg4 cfg5; do
  timeout 300 python bench.py --workload $w --no-also > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err
done
timeout 300 python bench.py --deterministic --no-also --no-cpu-baseline > gpurun_out/bench_cfg3b_deterministic.json 2>/dev/null
tail -c 600 gpurun_out/bench_default.json
