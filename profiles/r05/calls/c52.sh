# round 5, call 52: the tree's base address pinned in the mesh walks (it was re-fetched from the argument block in front of every node fetch): parity, A/B
timeout 1800 python -m pytest tests/test_gpu_render_parity.py tests/test_gpu_switch_matrix.py -m gpu -q -x --timeout=900 > gpurun_out/c52_pytest.log 2>&1; tail -1 gpurun_out/c52_pytest.log
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-72s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
for rep in 1 2; do
for a in "--workload big-soup --samples 64" "--workload big-mesh --samples 64" "--workload cows" "--workload mirror" "--workload mirror --traversal hier" "--workload big-soup --samples 64 --traversal hier"; do
  bash profiles/r05/with_objs.sh "1=build/diag/m1_before.o 8=build/diag/m8_before.o" python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "base re-fetched $a"
  python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "base pinned $a"
done; done > gpurun_out/c52_base_pinned.txt 2>&1
cat gpurun_out/c52_base_pinned.txt
