# round 5, call 75: the same in the k-d instantiations with meshes (modes 2, 9): parity, A/B
OBJ="9=build/diag/m9_slowblock.o 2=build/diag/m2_slowblock.o"
bash profiles/r05/with_objs.sh "$OBJ" timeout 900 python -m pytest tests/test_gpu_render_parity.py tests/test_gpu_config_sizes.py -m gpu -q -x -k "kd" --timeout=900 > gpurun_out/c75_pytest.log 2>&1; tail -1 gpurun_out/c75_pytest.log
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-72s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
for rep in 1 2; do for a in "--workload mirror --traversal kd" "--workload cows --traversal kd" "--workload big-soup --traversal kd" "--workload aquarium --traversal kd"; do
  python3 bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 $a 2>/dev/null | line "shipped $a"
  bash profiles/r05/with_objs.sh "$OBJ" python3 bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 $a 2>/dev/null | line "general case in one block $a"
done; done > gpurun_out/c75_kd_slow_block_mesh.txt 2>&1
cat gpurun_out/c75_kd_slow_block_mesh.txt
