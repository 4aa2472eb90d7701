# round 5, call 16: which of the k-d walk's changes costs: A/B objects of mode 7
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-58s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
B="python3 bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 --workload big-scene --traversal kd"
for n in kd_ef2 kd_nooct kd_nopopc kd_nobranch kd_none; do
bash profiles/r05/with_objs.sh "7=build/diag/$n.o" $B 2>/dev/null | line "kd big-scene, $n"
done
