# round 5, call 27: the quadratic's larger root computed only where the smaller one is not the answer (one f64 division fewer in most primitive tests) - headline A/B
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-58s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
B="python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 --workload big-scene"
for n in roots_both roots_lazy roots_both roots_lazy; do
bash profiles/r05/with_objs.sh "3=build/diag/$n.o" $B 2>/dev/null | line "big-scene, $n"
done
