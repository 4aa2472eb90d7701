# round 5, call 64: below k-d leaves: the octant instantiations (for big meshes) against one per-lane instantiation - what the extra code costs the scenes that never take it
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-72s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
OBJ="9=build/diag/m9_nooct.o 2=build/diag/m2_nooct.o"
for rep in 1 2; do
for a in "--workload mirror --traversal kd" "--workload cows --traversal kd" "--workload big-soup --traversal kd" "--workload big-soup --samples 64 --traversal kd"; do
  python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "with the octant instantiations $a"
  bash profiles/r05/with_objs.sh "$OBJ" python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "per-lane form only $a"
done; done > gpurun_out/c64_below_kd.txt 2>&1
cat gpurun_out/c64_below_kd.txt
