# SQ / LDS counters of the forward + adjoint kernel alone (tools/probe_early.py as the workload); separate --pmc passes.
#   bash tools/profile_early.sh <label> [env assignments for the workload ...]
R=$GRAFT_REPO_ROOT
label=$1; shift
cd /tmp && export TMPDIR=/tmp
B="env $* python $R/tools/probe_early.py 26 20 $label"
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES -d /tmp/pe_a_$label -- $B > /tmp/pe_a.log 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES SQ_INST_LEVEL_VMEM -d /tmp/pe_b_$label -- $B > /tmp/pe_b.log 2>&1
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_IFETCH_LEVEL SQ_LEVEL_WAVES -d /tmp/pe_c_$label -- $B > /tmp/pe_c.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_ATOMIC_RETURN SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU SQ_BUSY_CYCLES -d /tmp/pe_d_$label -- $B > /tmp/pe_d.log 2>&1
cd $R
tail -n 1 /tmp/pe_a.log
python tools/rocprof_summary.py raw /tmp/pe_a_$label /tmp/pe_b_$label /tmp/pe_c_$label /tmp/pe_d_$label > gpurun_out/rocprof_sq_early_$label.txt
grep -A34 "k_bucket_pair_forward_adjoint\|k_page_partition" gpurun_out/rocprof_sq_early_$label.txt | grep -v "^--" | head -80
