set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-also --steps 5 --warmup 2 --profile-steps 2"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- $B > /tmp/kt.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_fetch -- $B > /tmp/f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_write -- $B > /tmp/w.log 2>&1
cd $R
python tools/rocprof_summary.py kernels /tmp/prof_kt > gpurun_out/rocprof_kernel_stats.txt
python tools/rocprof_summary.py pmc /tmp/prof_fetch /tmp/prof_write > gpurun_out/rocprof_pmc.txt
tail -3 /tmp/kt.log /tmp/f.log /tmp/w.log
head -30 gpurun_out/rocprof_kernel_stats.txt
