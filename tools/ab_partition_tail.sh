cd $GRAFT_REPO_ROOT
for v in base spread noactive both base; do
  cp build/variants/probe-$v.so enoki_amd/libenoki-hip-probe.so
  echo "== $v"
  EK_PG_TIMING=1 PROBE_N=8388608 PROBE_SHIFT=12 PROBE_SKIP_SKEW=1 python tools/probe_paged.py time 2>&1 | grep "directory=0\|epilogue\|wall clock" | cut -c1-330
done
for v in base both; do
  cp build/variants/probe-$v.so enoki_amd/libenoki-hip-probe.so
  echo "== $v 64Mi"
  EK_PG_TIMING=1 PROBE_SHIFT=12 PROBE_SKIP_SKEW=1 python tools/probe_paged.py time 2>&1 | grep "directory=0\|epilogue\|wall clock" | cut -c1-330
done
