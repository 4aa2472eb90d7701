# round 5, call 22: k-d walk, the refined reciprocals on demand (six registers fewer) against held for the walk; and 4 against 5 waves per SIMD with the recomputed top levels
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-58s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
B="python3 bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 --workload big-scene --traversal kd"
for n in kd_base kd_rcp; do
bash profiles/r05/with_objs.sh "7=build/diag/$n.o" $B 2>/dev/null | line "kd big-scene, $n"
PORTRAYER_KD_WAVES=4 bash profiles/r05/with_objs.sh "7=build/diag/$n.o" $B 2>/dev/null | line "kd big-scene, $n, 4 waves"
done
