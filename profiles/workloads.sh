#!/bin/bash
# Throughput of every workload / traversal with the current build (no CPU baseline).
for wl in big-scene mirror cows primitives; do for tr in flat kd; do
python bench.py --no-cpu-baseline --steps 2 --warmup 1 --workload $wl --traversal $tr 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']['per_ray']
print('%-12s %-4s %9.1f Mray/s %9.2f ms/frame  rays/frame %.3g  nodes/ray %.1f prim/ray %.2f tri/ray %.2f' % ('$wl','$tr', d['value'], d['ms_per_step'], d['config']['rays_per_frame'], r['inner_nodes'], r['primitive_tests'], r['triangle_tests']))"
done; done
