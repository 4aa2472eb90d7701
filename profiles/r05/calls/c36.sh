# round 5, call 36: big-soup, host-built (SAH, depth-first array) against device-built (PLOC, creation-order array) tree; leaf sizes
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-72s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
for cfg in "PORTRAYER_BUILD=device" "PORTRAYER_BUILD=host" "PORTRAYER_BUILD=device PORTRAYER_PLOC_RADIUS=32" "PORTRAYER_BUILD=device PORTRAYER_PLOC_RADIUS=64"; do
for a in "--workload big-soup --samples 64"; do
  env $cfg python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "$cfg $a"
done; done > gpurun_out/c36_soup_trees.txt 2>&1
cat gpurun_out/c36_soup_trees.txt
