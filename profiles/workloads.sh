#!/bin/bash
# Throughput of the headline workloads with the current build (no CPU baseline). usage: bash profiles/workloads.sh [extra bench args]
run() {
python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']['per_ray']
print('%-44s %9.1f Mray/s %9.2f ms/frame (kernel %8.2f)  rays/frame %.3g  nodes/ray %.1f prim/ray %.2f tri/ray %.2f' % ('$*', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['rays_per_frame'], r['inner_nodes'], r['primitive_tests'], r['triangle_tests']))"
}
EXTRA="$@"
run --workload big-scene $EXTRA
run --workload big-scene --traversal hier $EXTRA
run --workload big-scene --traversal kd $EXTRA
run --workload mirror $EXTRA
run --workload mirror --traversal hier $EXTRA
run --workload cows $EXTRA
run --workload aquarium $EXTRA
run --workload big-soup $EXTRA
run --workload big-mesh $EXTRA
