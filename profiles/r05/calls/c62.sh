# round 5, call 62: the k-d split's common case (no lane crosses, every lane on one side) settled in seven scalar instructions (modes 2, 7, 9): parity of the k-d cases, A/B
OBJ="7=build/diag/m7_fast.o 9=build/diag/m9_fast.o 2=build/diag/m2_fast.o"
bash profiles/r05/with_objs.sh "$OBJ" timeout 1500 python -m pytest tests/test_gpu_render_parity.py tests/test_gpu_config_sizes.py -m gpu -q -x -k "kd" --timeout=900 > gpurun_out/c62_pytest.log 2>&1; tail -1 gpurun_out/c62_pytest.log
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-72s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
for rep in 1 2; do
for a in "--workload big-scene --traversal kd" "--workload mirror --traversal kd" "--workload cows --traversal kd"; do
  python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "shipped $a"
  bash profiles/r05/with_objs.sh "$OBJ" python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "common case first $a"
done; done > gpurun_out/c62_kd_fast.txt 2>&1
cat gpurun_out/c62_kd_fast.txt
