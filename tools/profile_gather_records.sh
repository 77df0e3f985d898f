# L2 request / miss counters of a struct gather (C = 3 tables of 8 Mi f32, 16 Mi lookups): plain kernel vs staged records.
# Run on the GPU box from the repository root; writes gpurun_out/rocprof_gather_records.txt
set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/tools/probe_gather_records.py --one"
timeout 200 rocprofv3 --pmc TCC_REQ_sum TCC_MISS_sum -d /tmp/prof_gra -- $B > /tmp/a.log 2>&1
timeout 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum -d /tmp/prof_grb -- $B > /tmp/b.log 2>&1
cd $R
for f in /tmp/a.log /tmp/b.log; do tail -n 3 $f; done
python tools/rocprof_summary.py raw /tmp/prof_gra /tmp/prof_grb > gpurun_out/rocprof_gather_records.txt
head -40 gpurun_out/rocprof_gather_records.txt
