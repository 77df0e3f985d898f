R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --workload cfg4 --no-cpu-baseline --no-also --steps 3 --warmup 1 --profile-steps 1"
timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof4_fetch -- $B > /tmp/f4.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof4_write -- $B > /tmp/w4.log 2>&1
cd $R
python tools/rocprof_summary.py pmc /tmp/prof4_fetch /tmp/prof4_write > gpurun_out/rocprof_pmc_cfg4.txt
