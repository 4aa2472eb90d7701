# round 5, call 79: k-d nodes whose cull leaves no lane skip the split's evaluation (mode 7): parity, A/B
bash profiles/r05/with_objs.sh "7=build/diag/m7_skip.o" timeout 900 python -m pytest tests/test_gpu_render_parity.py tests/test_gpu_config_sizes.py -m gpu -q -x -k "kd" --timeout=900 > gpurun_out/c79_pytest.log 2>&1; tail -1 gpurun_out/c79_pytest.log
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-72s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
for rep in 1 2 3; do
  python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 --workload big-scene --traversal kd 2>/dev/null | line "shipped big-scene kd"
  bash profiles/r05/with_objs.sh "7=build/diag/m7_skip.o" python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 --workload big-scene --traversal kd 2>/dev/null | line "culled nodes skipped big-scene kd"
done > gpurun_out/c79_kd_skip.txt 2>&1
cat gpurun_out/c79_kd_skip.txt
