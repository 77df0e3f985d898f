cp enoki_amd/libenoki-hip.so /tmp/prod.so
for r in 1 2; do
cp build/variants/libenoki-hip-firstwave.so enoki_amd/libenoki-hip.so
for w in cfg2 cfg3a; do python bench.py --workload $w --no-cpu-baseline --no-also --steps 40 --warmup 5 2>/dev/null | python tools/bench_line.py - | head -1 | sed "s/^/firstwave $w /"; done
cp /tmp/prod.so enoki_amd/libenoki-hip.so
for w in cfg2 cfg3a; do python bench.py --workload $w --no-cpu-baseline --no-also --steps 40 --warmup 5 2>/dev/null | python tools/bench_line.py - | head -1 | sed "s/^/product   $w /"; done
done
python tools/probe_second_wave.py 26
