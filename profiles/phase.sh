#!/bin/bash
# Where do the wave cycles go? Builds with -DPT_PHASE_TIMING and prints the share of the stats kernel's
# cycles spent in pt_lane_advance (n_tri), in pt_trace (n_bbox) and in the work hand-out (kd_plane_miss).
# usage: bash profiles/phase.sh "<bench args>"
rm -f portrayer_amd/libportrayer_hip.so portrayer_amd/csrc/pt_api.o
make -s -j2 portrayer_amd/libportrayer_hip.so EXTRA_HIPFLAGS="-DPT_PHASE_TIMING" > /dev/null 2>&1 || { echo build failed; exit 1; }
PT_DUMP_COUNTERS=1 python bench.py --no-cpu-baseline --steps 1 --warmup 0 $1 2>&1 | grep "^counters" | python -c "
import sys, json
d = json.loads(sys.stdin.read().split(' ', 1)[1])
a, t, w = d['n_tri'], d['n_bbox'], d['kd_plane_miss']
tot = a + t + w
print('advance %.1f %%  trace %.1f %%  hand-out %.1f %%  (wave-cycles x64: %.3g)' % (100*a/tot, 100*t/tot, 100*w/tot, tot))"
rm -f portrayer_amd/libportrayer_hip.so portrayer_amd/csrc/pt_api.o
make -s -j2 portrayer_amd/libportrayer_hip.so > /dev/null 2>&1
