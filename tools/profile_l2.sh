set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-also --steps 3 --warmup 1 --profile-steps 1"
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d /tmp/prof_l2a -- $B > /tmp/a.log 2>&1
timeout 300 rocprofv3 --pmc TCC_REQ_sum TCC_READ_sum -d /tmp/prof_l2b -- $B > /tmp/b.log 2>&1
timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum -d /tmp/prof_l2c -- $B > /tmp/c.log 2>&1
cd $R
for f in /tmp/a.log /tmp/b.log /tmp/c.log; do tail -n 2 $f; done
python tools/rocprof_summary.py raw /tmp/prof_l2a /tmp/prof_l2b /tmp/prof_l2c > gpurun_out/rocprof_l2.txt
cat gpurun_out/rocprof_l2.txt | head -80
