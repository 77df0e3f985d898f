# rocprofv3 evidence for round 4 (run on the GPU box: bash tools/profile_r04.sh); summaries land in gpurun_out/ and are copied to
# profiles/*_r04.txt.  Counter passes are separate runs (no tracing domains besides the kernel trace).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-also --steps 5 --warmup 2 --profile-steps 2"
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- $B > /tmp/kt.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_fetch -- $B > /tmp/f.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_write -- $B > /tmp/w.log 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU -d /tmp/prof_sqa -- $B > /tmp/c.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d /tmp/prof_sqb -- $B > /tmp/d.log 2>&1
# cfg4 per pixel in bucket order and the fused cfg5: kernel trace + HBM traffic
for w in cfg4_bucketed cfg5; do
  W="python $R/bench.py --workload $w --no-cpu-baseline --no-also --steps 5 --warmup 2 --profile-steps 2"
  timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt_$w -- $W > /tmp/kt_$w.log 2>&1
  timeout 200 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_fetch_$w -- $W > /tmp/f_$w.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_write_$w -- $W > /tmp/w_$w.log 2>&1
done
# the sliced table (K = 16 Mi) and the 8 Mi-element shard of an 8-way split: kernel trace only
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt_k16 -- python $R/bench.py --workload cfg3b_K16Mi --no-cpu-baseline --no-also --steps 5 --warmup 2 --profile-steps 2 > /tmp/kt_k16.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt_8mi -- python $R/bench.py --n 8388608 --no-cpu-baseline --no-also --steps 20 --warmup 2 --profile-steps 2 > /tmp/kt_8mi.log 2>&1
cd $R
python tools/rocprof_summary.py kernels /tmp/prof_kt_k16 > gpurun_out/rocprof_kernel_stats_cfg3b_K16Mi_r04.txt
python tools/rocprof_summary.py kernels /tmp/prof_kt_8mi > gpurun_out/rocprof_kernel_stats_cfg3b_8Mi_r04.txt
python tools/rocprof_summary.py kernels /tmp/prof_kt > gpurun_out/rocprof_kernel_stats_r04.txt
python tools/rocprof_summary.py pmc /tmp/prof_fetch /tmp/prof_write > gpurun_out/rocprof_pmc_r04.txt
python tools/rocprof_summary.py raw /tmp/prof_sqa /tmp/prof_sqb > gpurun_out/rocprof_sq_r04.txt
for w in cfg4_bucketed cfg5; do
  python tools/rocprof_summary.py kernels /tmp/prof_kt_$w > gpurun_out/rocprof_kernel_stats_${w}_r04.txt
  python tools/rocprof_summary.py pmc /tmp/prof_fetch_$w /tmp/prof_write_$w > gpurun_out/rocprof_pmc_${w}_r04.txt
done
head -14 gpurun_out/rocprof_kernel_stats_r04.txt
# the two bandwidth-bound reference points (BASELINE configs[1] and the leaf-only tape): kernel trace + HBM traffic
cd /tmp
for w in cfg3a cfg2; do
  W="python $R/bench.py --workload $w --no-cpu-baseline --no-also --steps 5 --warmup 2 --profile-steps 2"
  timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt_$w -- $W > /tmp/kt_$w.log 2>&1
  timeout 200 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_fetch_$w -- $W > /tmp/f_$w.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_write_$w -- $W > /tmp/w_$w.log 2>&1
done
cd $R
for w in cfg3a cfg2; do
  python tools/rocprof_summary.py kernels /tmp/prof_kt_$w > gpurun_out/rocprof_kernel_stats_${w}_r04.txt
  python tools/rocprof_summary.py pmc /tmp/prof_fetch_$w /tmp/prof_write_$w > gpurun_out/rocprof_pmc_${w}_r04.txt
done
