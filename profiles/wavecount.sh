#!/bin/bash
# Lane occupancy inside the tree walk of a mesh-free scene (big-scene): builds with -DPT_WAVE_COUNTS, which
# counts wavefront passes through the inner-node step (in n_tri) and the leaf step (in n_bbox).
rm -f portrayer_amd/libportrayer_hip.so portrayer_amd/csrc/pt_api.o
make -s -j2 portrayer_amd/libportrayer_hip.so EXTRA_HIPFLAGS="-DPT_WAVE_COUNTS" > /dev/null 2>&1 || { echo build failed; exit 1; }
PT_DUMP_COUNTERS=1 python bench.py --no-cpu-baseline --steps 1 --warmup 0 $1 2>&1 | grep "^counters" | python -c "
import sys, json
d = json.loads(sys.stdin.read().split(' ', 1)[1])
rays = d['primary'] + d['shadow'] + d['reflect'] + d['refract']
print('inner-node steps: %.3g lane, %.3g wave -> %.1f of 64 lanes; per 64 rays %.1f wave steps' % (d['n_inner'], d['n_tri'], d['n_inner'] / d['n_tri'], d['n_tri'] / (rays / 64)))
print('leaf steps      : %.3g lane, %.3g wave -> %.1f of 64 lanes; per 64 rays %.1f wave steps' % (d['n_leaf'], d['n_bbox'], d['n_leaf'] / d['n_bbox'], d['n_bbox'] / (rays / 64)))"
rm -f portrayer_amd/libportrayer_hip.so portrayer_amd/csrc/pt_api.o
make -s -j2 portrayer_amd/libportrayer_hip.so > /dev/null 2>&1
