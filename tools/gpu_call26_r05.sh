# round 5: class weights with hysteresis (dealt from 4 %, equal again below 2 %): the state on this box, the kernels with / without
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/probe_xcd_balance.py 2>&1 | tail -4 | tee gpurun_out/probe_xcd_state.txt
