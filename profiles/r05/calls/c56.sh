# round 5, call 56: what the instance walk's watchdog and the push's overflow test cost (A/B objects that leave them out; never shipped)
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-72s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
for rep in 1 2; do
for a in "--workload big-soup --samples 64" "--workload mirror" "--workload cows"; do
  python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "shipped $a"
  bash profiles/r05/with_objs.sh "1=build/diag/m1_nowatch.o" python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "no watchdog in the instance walk $a"
  bash profiles/r05/with_objs.sh "1=build/diag/m1_nopushcheck.o" python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "no overflow test at the push $a"
done; done > gpurun_out/c56_safety_costs.txt 2>&1
cat gpurun_out/c56_safety_costs.txt
