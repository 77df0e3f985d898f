# SQ / LDS counters of the headline step's kernels (where do the waves of the scatter pipeline wait?)
set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-also --steps 3 --warmup 1 --profile-steps 1 --eager"
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS -d /tmp/prof_sqa -- $B > /tmp/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d /tmp/prof_sqb -- $B > /tmp/b.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU -d /tmp/prof_sqc -- $B > /tmp/c.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d /tmp/prof_sqd -- $B > /tmp/d.log 2>&1
cd $R
for f in /tmp/a.log /tmp/b.log /tmp/c.log /tmp/d.log; do tail -n 2 $f; done
python tools/rocprof_summary.py raw /tmp/prof_sqa /tmp/prof_sqb /tmp/prof_sqc /tmp/prof_sqd > gpurun_out/rocprof_sq.txt
grep -A18 "k_bin_partition\|k_bin_accumulate" gpurun_out/rocprof_sq.txt | head -80
