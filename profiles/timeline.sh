#!/bin/bash
# When do wavefronts end? From a -DPT_TIMELINE build made beforehand with
#   make variant NAME=timeline EXTRA_HIPFLAGS=-DPT_TIMELINE
# the counting launch reports wavefront lifetimes, the launch's span and the longest work item (100 MHz wall clock).
# usage (GPU box, repo root): bash profiles/timeline.sh "<bench args>" ["<bench args>" ...]
cp portrayer_amd/libportrayer_hip.so /tmp/libportrayer_hip.so.keep
cp build/variants/timeline/libportrayer_hip.so portrayer_amd/libportrayer_hip.so
for BARGS in "$@"; do
PT_DUMP_COUNTERS=1 python3 bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 0 $BARGS 2>&1 | grep "^counters" | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().split(' ', 1)[1]); g = d['diag']
span = g[5] - ((~g[4]) & 0xFFFFFFFFFFFFFFFF)
tail = (g[5] - ((~g[6]) & 0xFFFFFFFFFFFFFFFF)) if g[6] else 0
print('%-50s launch span %.3f ms, %d wavefronts, mean lifetime %.3f ms = %.0f %% of the span, longest item %.3f ms, first wavefront done %.3f ms before the last, start -> first item %.4f ms' % ('$BARGS', span / 1e5, g[1], g[0] / g[1] / 1e5, 100.0 * g[0] / g[1] / span, g[2] / 1e5, tail / 1e5, g[3] / max(g[1], 1) / 1e5))
"
done
cp /tmp/libportrayer_hip.so.keep portrayer_amd/libportrayer_hip.so
