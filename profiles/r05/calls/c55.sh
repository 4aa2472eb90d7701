# round 5, call 55: the tree's base pinned at the scene level of the mesh walks too (modes 1, 4, 5, 8): parity, A/B
timeout 1500 python -m pytest tests/test_gpu_render_parity.py tests/test_gpu_textures.py -m gpu -q -x --timeout=900 > gpurun_out/c55_pytest.log 2>&1; tail -1 gpurun_out/c55_pytest.log
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-72s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
B="1=build/diag/m1_before.o 4=build/diag/m4_before.o 5=build/diag/m5_before.o 8=build/diag/m8_before.o"
for rep in 1 2; do
for a in "--workload cows" "--workload mirror" "--workload mirror --traversal hier" "--workload aquarium" "--workload aquarium --traversal hier" "--workload big-mesh --samples 64"; do
  bash profiles/r05/with_objs.sh "$B" python3 bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 $a 2>/dev/null | line "before $a"
  python3 bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 $a 2>/dev/null | line "scene-level base pinned $a"
done; done > gpurun_out/c55_scene_base.txt 2>&1
cat gpurun_out/c55_scene_base.txt
