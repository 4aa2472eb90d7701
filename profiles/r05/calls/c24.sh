# round 5, call 24: the dielectric workload's interpreter kernel compiled for 2 waves per SIMD (256 registers: no spills) against the shipped 3 (168 registers, 94 spilled)
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-64s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
B="python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload aquarium"
$B 2>/dev/null | line "aquarium, shipped (3 waves)"
bash profiles/r05/with_objs.sh "4=build/diag/interp2w.o" $B 2>/dev/null | line "aquarium, interpreter at 2 waves"
PORTRAYER_LDS_BUDGET_KB=80 bash profiles/r05/with_objs.sh "4=build/diag/interp2w.o" $B 2>/dev/null | line "aquarium, interpreter at 2 waves, 80 KB of LDS per block"
