# round 5, call 29: the k-d walk with the guarded bookkeeping and the pinned cull flags: parity of every k-d test and the switch matrix, timings
timeout 1500 python3 -m pytest tests -m gpu -q -k "kd or KD or switch or headline or config_size or timed_size or fuzz" > gpurun_out/c29_tests.txt 2>&1; grep -h "passed\|failed" gpurun_out/c29_tests.txt | tail -1
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-58s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
for a in "--workload big-scene --traversal kd" "--workload mirror --traversal kd" "--workload cows --traversal kd" "--workload big-soup --traversal kd"; do
  python3 bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 $a 2>/dev/null | line "$a"
done > gpurun_out/c29_kd.txt 2>&1; cat gpurun_out/c29_kd.txt
timeout 900 python3 tests/fuzz_gpu_parity.py 210000 40 > gpurun_out/c29_fuzz.log 2>&1; tail -1 gpurun_out/c29_fuzz.log
