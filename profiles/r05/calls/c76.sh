# round 5, call 76: the k-d split with both of its cases in hand-written blocks as the default: k-d parity, switch matrix, k-d workloads, fuzz
timeout 1800 python -m pytest tests/test_gpu_render_parity.py tests/test_gpu_config_sizes.py tests/test_gpu_switch_matrix.py tests/test_gpu_timed_sizes.py -m gpu -q -x --timeout=900 > gpurun_out/c76_pytest.log 2>&1; tail -1 gpurun_out/c76_pytest.log
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-72s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
for a in "--workload big-scene --traversal kd" "--workload mirror --traversal kd" "--workload cows --traversal kd" "--workload big-soup --traversal kd" "--workload aquarium --traversal kd"; do
  python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "$a"
done > gpurun_out/c76_kd.txt 2>&1
cat gpurun_out/c76_kd.txt
timeout 1200 python3 tests/fuzz_gpu_parity.py 295000 80 > gpurun_out/c76_fuzz.log 2>&1; tail -1 gpurun_out/c76_fuzz.log
