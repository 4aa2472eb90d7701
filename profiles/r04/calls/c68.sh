# round 4, call 68: how the k-d semantics' work items are handed out, on the wave-uniform walk (round 3 measured batches 1 % ahead for the per-lane walk)
run() { name=$1; shift
  env "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-44s %9.1f Mray/s %8.3f ms' % ('$name', d['value'], d['ms_per_step']))"
}
B="python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --traversal kd"
for w in big-scene mirror cows; do
run "$w kd default (batches)" X=1 $B --workload $w
run "$w kd FINE_QUEUES=16" PORTRAYER_FINE_QUEUES=16 $B --workload $w
run "$w kd FINE_QUEUES=32" PORTRAYER_FINE_QUEUES=32 $B --workload $w
run "$w kd BATCH_MAX=2" PORTRAYER_BATCH_MAX=2 $B --workload $w
done > gpurun_out/c68_kd_handout.txt 2>&1
cat gpurun_out/c68_kd_handout.txt
