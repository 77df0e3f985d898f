# rocprofv3 evidence for round 6 (run on the GPU box: bash tools/profile_r06.sh [quick]); summaries land in gpurun_out/ and are
# copied to profiles/*_r06.txt.  Counter passes are separate runs (no tracing domains besides the kernel trace).
# The headline trace is taken with the DRIVER's protocol (--steps 20 --warmup 5) behind bench.py's own pre-warm (0.6 s of the
# headline step): the average kernel durations are those of the state the timed steps run in, and `# step_sum_us` (kernels of one
# step back to back) is what bench.py's trace_check compares its ms_per_step with.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-also --steps 20 --warmup 5 --profile-steps 2"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- $B > /tmp/kt.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_fetch -- $B --pre-warm-s 0.05 > /tmp/f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_write -- $B --pre-warm-s 0.05 > /tmp/w.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU -d /tmp/prof_sqa -- $B --pre-warm-s 0.05 > /tmp/c.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d /tmp/prof_sqb -- $B --pre-warm-s 0.05 > /tmp/d.log 2>&1
cd $R
python tools/rocprof_summary.py kernels /tmp/prof_kt > gpurun_out/rocprof_kernel_stats_r06.txt
python tools/rocprof_summary.py pmc /tmp/prof_fetch /tmp/prof_write > gpurun_out/rocprof_pmc_r06.txt
python tools/rocprof_summary.py raw /tmp/prof_sqa /tmp/prof_sqb > gpurun_out/rocprof_sq_r06.txt
head -14 gpurun_out/rocprof_kernel_stats_r06.txt; grep "^# step" gpurun_out/rocprof_kernel_stats_r06.txt
[ "$1" = quick ] && exit 0
cd /tmp
# BASELINE configs[1] as one pass (cfg2), the leaf-only tape (cfg3a), the 8 Mi-element shard of an 8-way split, cfg4 per pixel
for w in cfg2 cfg3a cfg4_bucketed; do
  W="python $R/bench.py --workload $w --no-cpu-baseline --no-also --steps 20 --warmup 5 --profile-steps 2 --pre-warm-s 0.2"
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt_$w -- $W > /tmp/kt_$w.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_fetch_$w -- $W > /tmp/f_$w.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_write_$w -- $W > /tmp/w_$w.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt_8mi -- python $R/bench.py --n 8388608 --no-cpu-baseline --no-also --steps 50 --warmup 5 --profile-steps 2 --pre-warm-s 0.2 > /tmp/kt_8mi.log 2>&1
cd $R
for w in cfg2 cfg3a cfg4_bucketed; do
  python tools/rocprof_summary.py kernels /tmp/prof_kt_$w > gpurun_out/rocprof_kernel_stats_${w}_r06.txt
  python tools/rocprof_summary.py pmc /tmp/prof_fetch_$w /tmp/prof_write_$w > gpurun_out/rocprof_pmc_${w}_r06.txt
done
python tools/rocprof_summary.py kernels /tmp/prof_kt_8mi > gpurun_out/rocprof_kernel_stats_cfg3b_8Mi_r06.txt
