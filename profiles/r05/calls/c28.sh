# round 5, call 28: k-d split bookkeeping behind wave-uniform guards (nothing pushed: nothing to write, no codes / bounds to select) - A/B, alternating
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-58s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
B="python3 bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 --workload big-scene --traversal kd"
for n in ${NAMES:-kd_plain kd_guards kd_plain kd_guards}; do
bash profiles/r05/with_objs.sh "7=build/diag/$n.o" $B 2>/dev/null | line "kd big-scene, $n"
done
