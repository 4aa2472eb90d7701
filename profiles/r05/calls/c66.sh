# round 5, call 66: flat / hier mesh scenes with ONE instantiation of the instance walk (per-lane form) against the nine: what the octant copies cost short walks
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-72s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
OBJ="1=build/diag/m1_one.o 8=build/diag/m8_one.o"
for rep in 1 2; do
for a in "--workload mirror" "--workload cows" "--workload mirror --traversal hier" "--workload big-mesh --samples 64" "--workload big-soup --samples 64"; do
  python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "nine instantiations $a"
  bash profiles/r05/with_objs.sh "$OBJ" python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "one $a"
done; done > gpurun_out/c66_one_instantiation.txt 2>&1
cat gpurun_out/c66_one_instantiation.txt
