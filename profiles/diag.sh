#!/bin/bash
# Lane occupancy of the render kernel's phases, from a -DPT_DIAG build made beforehand with
#   make variant NAME=diag EXTRA_HIPFLAGS=-DPT_DIAG
# usage (GPU box, repo root): bash profiles/diag.sh "<bench args>" ["<bench args>" ...]
cp portrayer_amd/libportrayer_hip.so /tmp/libportrayer_hip.so.keep
cp build/variants/diag/libportrayer_hip.so portrayer_amd/libportrayer_hip.so
for BARGS in "$@"; do
PT_DUMP_COUNTERS=1 python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 $BARGS 2>&1 | grep "^counters" | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().split(' ', 1)[1]); g = d['diag']
rays = d['primary'] + d['shadow'] + d['reflect'] + d['refract']
def r(a, b): return a / b if b else float('nan')
print('%-60s rays %.3g (shadow %.0f %%, secondary %.0f %%)' % ('$BARGS', rays, 100 * d['shadow'] / rays, 100 * (d['reflect'] + d['refract']) / rays))
print('   trace calls : %5.1f of 64 lanes carry a ray (any-hit calls: %5.1f); %.2f wave calls per 64 rays' % (r(g[1], g[0]), r(g[7], g[6]), r(g[0], rays / 64)))
print('   interpreter : %5.1f of 64 lanes active' % r(g[3], g[2]))
print('   inner steps : %5.1f of 64 lanes; %.1f per ray, %.1f wave steps per trace call' % (r(d['n_inner'], g[4]), r(d['n_inner'], rays), r(g[4], g[0])))
print('   leaf steps  : %5.1f of 64 lanes; %.2f per ray, %.1f wave steps per trace call' % (r(d['n_leaf'], g[5]), r(d['n_leaf'], rays), r(g[5], g[0])))
"
done
cp /tmp/libportrayer_hip.so.keep portrayer_amd/libportrayer_hip.so
