#!/bin/bash
# 3 vs 4 waves per SIMD on the workloads whose kernels have both instantiations.
for wl in "big-soup" "big-mesh" "cows" "mirror" "aquarium" "big-scene --traversal kd" "big-soup --traversal hier" "mirror --traversal kd"; do for wv in 3 4; do
PORTRAYER_WAVES=$wv python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload $wl 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-32s waves %s %9.1f Mray/s %9.2f ms/frame' % ('$wl', '$wv', d['value'], d['ms_per_step']))"
done; done
