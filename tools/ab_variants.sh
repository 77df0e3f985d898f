# A/B of build/variants/libenoki-hip-<name>.so against the product library inside ONE gpurun call (tools/build_variant.py makes them):
#   bash tools/ab_variants.sh <name> [<name> ...]      -- forward + adjoint kernel alone, 64 Mi and 8 Mi elements (tools/probe_early.py)
cd $GRAFT_REPO_ROOT
cp enoki_amd/libenoki-hip.so /tmp/prod.so
for r in 1 2; do
  python tools/probe_early.py 26 20 product
  for v in "$@"; do
    cp build/variants/libenoki-hip-$v.so enoki_amd/libenoki-hip.so
    python tools/probe_early.py 26 20 $v
    cp /tmp/prod.so enoki_amd/libenoki-hip.so
  done
done
python tools/probe_early.py 23 20 product-8Mi
for v in "$@"; do
  cp build/variants/libenoki-hip-$v.so enoki_amd/libenoki-hip.so
  python tools/probe_early.py 23 20 $v-8Mi
  cp /tmp/prod.so enoki_amd/libenoki-hip.so
done
