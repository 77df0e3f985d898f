for r in 1 2; do
PROBE_HINTS=1 python tools/probe_early.py 26 20 locks-8Ki
python tools/probe_early.py 26 20 fixed-4Ki
done
PROBE_HINTS=1 python tools/probe_early.py 23 20 locks-8Ki-8Mi
python tools/probe_early.py 23 20 fixed-4Ki-8Mi
