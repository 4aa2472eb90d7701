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
# interpreter kernel: diag[2] = interpreter cycles, diag[0] = walk cycles; straight-line kernel: diag[2] = whole items, diag[0] = walks inside them
line = d.get('kernel_variant', 16) & 16 == 0
tot = g[2] if line else g[0] + g[2]
print('%-50s outside the walks %4.1f %%  walks %4.1f %% (leaf tests %4.1f %% of all, tree steps %4.1f %%)   %.0f wave cycles per walk' % ('$BARGS', 100 * (tot - g[0]) / tot, 100 * g[0] / tot, 100 * g[5] / tot, 100 * (g[0] - g[5]) / tot, g[0] / max(g[1], 1)))
if line and g[3]: print('%-50s   outside: primary ray %4.1f %%, surface of the hit %4.1f %%, light / shadow-ray set-up %4.1f %%, light term %4.1f %%, rest (work queue, chunk sums, finish) %4.1f %% of all' % ('', 100 * g[3] / tot, 100 * g[4] / tot, 100 * g[6] / tot, 100 * g[7] / tot, 100 * (tot - g[0] - g[3] - g[4] - g[6] - g[7]) / tot))"
done
cp /tmp/libportrayer_hip.so.keep portrayer_amd/libportrayer_hip.so
