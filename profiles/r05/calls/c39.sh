# round 5, call 39: the mesh walk's step loop without the vmcnt waits the 5-wave build carries (one wait in front of the descend instead): A/B, alternating
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-72s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
for rep in 1 2; do
for a in "--workload big-soup --samples 64" "--workload big-mesh --samples 64"; do
  python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "shipped $a"
  bash profiles/r05/with_objs.sh "1=build/diag/m1_wait_hoist.o" python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "wait hoisted $a"
done; done > gpurun_out/c39_wait_hoist.txt 2>&1
cat gpurun_out/c39_wait_hoist.txt
