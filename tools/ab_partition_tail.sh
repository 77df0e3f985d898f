# The tail of k_page_partition alone (tools/probe_paged.py, probe libraries built with -DEK_PG_TIMING): variants are copied from
# build/variants/probe-<name>.so (ENOKI_PROBE_DEFINES="-DEK_PG_TIMING -DEK_PG_REPLICAS=1" python -c 'from enoki_amd import _build as B; B.build_probe(force=True)',
# then cp enoki_amd/libenoki-hip-probe.so build/variants/probe-<name>.so).  Round 6 ran base / spread / noactive / both, where spread was a measurement
# variant of what became kPgReplicas.
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
