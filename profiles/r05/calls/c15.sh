# round 5, call 15: the k-d walk's culls from (entering, leaving) plane pairs per direction octant - timing of the mode-7 object against the shipped one
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-58s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
B="python3 bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 --workload big-scene --traversal kd"
bash profiles/r05/with_objs.sh "7=build/diag/$1.o" $B 2>/dev/null | line "kd big-scene, $1"
PORTRAYER_KD_WAVES=4 bash profiles/r05/with_objs.sh "7=build/diag/$1.o" $B 2>/dev/null | line "kd big-scene, $1, 4 waves"
