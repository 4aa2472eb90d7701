# round 5, call 26: the k-d walk's one-child fast path (no lane straddles, one side only: no push, no mask selects) against the general bookkeeping at every split
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-58s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
B="python3 bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 --workload big-scene --traversal kd"
for n in kd_noonechild kd_onechild; do
bash profiles/r05/with_objs.sh "7=build/diag/$n.o" $B 2>/dev/null | line "kd big-scene, $n"
done
