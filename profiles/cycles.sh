#!/bin/bash
# Where a wavefront's cycles go: interpreter / walk (of which: leaf tests), from a -DPT_CYCLES build made beforehand with
#   make variant NAME=cycles EXTRA_HIPFLAGS=-DPT_CYCLES
# (the leaf split is instrumented in the mesh-free two-child walk only). usage: bash profiles/cycles.sh "<bench args>" ...
cp portrayer_amd/libportrayer_hip.so /tmp/libportrayer_hip.so.keep
cp build/variants/cycles/libportrayer_hip.so portrayer_amd/libportrayer_hip.so
for BARGS in "$@"; do
PT_DUMP_COUNTERS=1 python3 bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 0 $BARGS 2>&1 | grep "^counters" | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().split(' ', 1)[1]); g = d['diag']
tot = g[0] + g[2]
print('%-50s interpreter %4.1f %%  walk %4.1f %% (leaf tests %4.1f %% of all)   %.0f cycles per loop iteration' % ('$BARGS', 100 * g[2] / tot, 100 * g[0] / tot, 100 * g[5] / tot, tot / max(g[1], 1)))"
done
cp /tmp/libportrayer_hip.so.keep portrayer_amd/libportrayer_hip.so
