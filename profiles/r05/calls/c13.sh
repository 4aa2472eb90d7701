# round 5, call 13: what do the k-d walk's saved bounds in HBM cost? The same kernel with the top levels' bounds aliased onto one LDS slot (wrong pictures, timing only)
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-58s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
B="python3 bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 --workload big-scene --traversal kd"
$B 2>/dev/null | line "kd big-scene, shipped"
bash profiles/r05/with_objs.sh "7=build/diag/kd_savfake.o" $B 2>/dev/null | line "kd big-scene, bounds of the top levels aliased in LDS"
PORTRAYER_KD_WAVES=4 $B 2>/dev/null | line "kd big-scene, shipped, 4 waves"
