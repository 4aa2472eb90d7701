# round 5, call 42: the tree step with the common case in six scalar instructions (vote / push / pop behind branches), modes 1 and 3: parity, then A/B
bash profiles/r05/with_objs.sh "1=build/diag/m1_branchy.o 3=build/diag/m3_branchy.o" timeout 1200 python3 -m pytest tests/test_gpu_render_parity.py -q -m gpu -x -k "example_matches or device_built or synthetic or random" > gpurun_out/c42_tests.txt 2>&1; tail -1 gpurun_out/c42_tests.txt
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-72s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
for rep in 1 2; do
for a in "--workload big-scene" "--workload big-soup --samples 64" "--workload big-mesh --samples 64" "--workload cows" "--workload mirror"; do
  python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "shipped $a"
  bash profiles/r05/with_objs.sh "1=build/diag/m1_branchy.o 3=build/diag/m3_branchy.o" python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "six-instruction step $a"
done; done > gpurun_out/c42_branchy.txt 2>&1
cat gpurun_out/c42_branchy.txt
