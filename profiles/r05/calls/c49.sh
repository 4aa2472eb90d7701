# round 5, call 49: the mesh walk below k-d leaves in the code of the instance's octant (modes 2, 9): parity of the k-d cases, then A/B
OBJ="2=build/diag/m2_oct.o 9=build/diag/m9_oct.o"
bash profiles/r05/with_objs.sh "$OBJ" timeout 1500 python -m pytest tests/test_gpu_render_parity.py tests/test_gpu_config_sizes.py tests/test_gpu_switch_matrix.py -m gpu -q -x -k "kd or switch" --timeout=900 > gpurun_out/c49_pytest.log 2>&1; tail -1 gpurun_out/c49_pytest.log
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-72s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
for rep in 1 2; do
for a in "--workload mirror --traversal kd" "--workload cows --traversal kd" "--workload big-soup --traversal kd"; do
  python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "per-lane form $a"
  bash profiles/r05/with_objs.sh "$OBJ" python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "octant form $a"
done; done > gpurun_out/c49_kd_oct.txt 2>&1
cat gpurun_out/c49_kd_oct.txt
