# round 5: class weights without a band (always dealt) -- do the classes' loop durations equalise, and does the kernel get shorter?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for band in 0 2621; do echo "band $band"; ENOKI_HIP_XCD_BAND=$band timeout 300 python tools/probe_xcd_balance.py 2>&1 | tail -4; done | tee gpurun_out/probe_xcd_state.txt
