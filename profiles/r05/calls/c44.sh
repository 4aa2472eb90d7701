# round 5, call 44: wave-level early exits in the triangle test (mode 1): parity subset, then A/B
bash profiles/r05/with_objs.sh "1=build/diag/m1_tri_exits.o" timeout 900 python3 -m pytest tests/test_gpu_render_parity.py -q -m gpu -x -k "example_matches or device_built or synthetic" > gpurun_out/c44_tests.txt 2>&1; tail -1 gpurun_out/c44_tests.txt
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-72s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
for rep in 1 2; do
for a in "--workload big-soup --samples 64" "--workload big-mesh --samples 64" "--workload cows" "--workload mirror"; do
  python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "shipped $a"
  bash profiles/r05/with_objs.sh "1=build/diag/m1_tri_exits.o" python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "triangle test with wave exits $a"
done; done > gpurun_out/c44_tri_exits.txt 2>&1
cat gpurun_out/c44_tri_exits.txt
