# round 5, call 37: the device-built tree's array re-laid depth first (host-side experiment): parity, then big-soup / big-mesh / cows
PORTRAYER_TREE_LAYOUT=dfs PORTRAYER_BUILD=device PORTRAYER_BUILD_MIN=16 timeout 900 python3 -m pytest tests/test_gpu_render_parity.py -q -m gpu -k "device_built or example_matches" > gpurun_out/c37_tests.txt 2>&1; tail -1 gpurun_out/c37_tests.txt
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-72s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
for cfg in "PORTRAYER_TREE_LAYOUT=none" "PORTRAYER_TREE_LAYOUT=dfs" "PORTRAYER_BUILD=host"; do
for a in "--workload big-soup --samples 64" "--workload big-soup --samples 16" "--workload big-mesh --samples 64"; do
  env $cfg python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "$cfg $a"
done; done > gpurun_out/c37_soup_layout.txt 2>&1
cat gpurun_out/c37_soup_layout.txt
