# rocprofv3 evidence for the headline step (run on the GPU box: bash tools/profile_r03.sh); summaries land in gpurun_out/
# and are copied to profiles/*_r03.txt.  Counter passes are separate runs (no tracing domains besides the kernel trace).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-also --steps 5 --warmup 2 --profile-steps 2 --eager"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- $B > /tmp/kt.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_fetch -- $B > /tmp/f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_write -- $B > /tmp/w.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d /tmp/prof_l2a -- $B > /tmp/a.log 2>&1
timeout 300 rocprofv3 --pmc TCC_REQ_sum TCC_READ_sum -d /tmp/prof_l2b -- $B > /tmp/b.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU -d /tmp/prof_sqa -- $B > /tmp/c.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d /tmp/prof_sqb -- $B > /tmp/d.log 2>&1
cd $R
python tools/rocprof_summary.py kernels /tmp/prof_kt > gpurun_out/rocprof_kernel_stats_r03.txt
python tools/rocprof_summary.py pmc /tmp/prof_fetch /tmp/prof_write > gpurun_out/rocprof_pmc_r03.txt
python tools/rocprof_summary.py raw /tmp/prof_l2a /tmp/prof_l2b > gpurun_out/rocprof_l2_r03.txt
python tools/rocprof_summary.py raw /tmp/prof_sqa /tmp/prof_sqb > gpurun_out/rocprof_sq_r03.txt
head -24 gpurun_out/rocprof_kernel_stats_r03.txt
