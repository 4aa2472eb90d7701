# round 5, call 45: the case-split step in the mesh walks below k-d leaves too: parity (k-d cases), k-d workloads
timeout 1500 python -m pytest tests/test_gpu_render_parity.py tests/test_gpu_config_sizes.py -m gpu -q -k "kd" --timeout=900 > gpurun_out/c45_pytest.log 2>&1; tail -1 gpurun_out/c45_pytest.log
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-72s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
for a in "--workload mirror --traversal kd" "--workload cows --traversal kd" "--workload big-soup --traversal kd" "--workload big-scene --traversal kd"; do
  python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "$a"
done > gpurun_out/c45_kd.txt 2>&1
cat gpurun_out/c45_kd.txt
