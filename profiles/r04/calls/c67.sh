# round 4, call 67: the number of work queues again on the final kernels
run() { name=$1; shift
  env "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-40s %9.1f Mray/s %8.3f ms' % ('$name', d['value'], d['ms_per_step']))"
}
B="python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1"
for q in 0 8 16 32 64; do
run "big-scene FINE_QUEUES=$q" PORTRAYER_FINE_QUEUES=$q $B --workload big-scene
run "big-soup x64 FINE_QUEUES=$q" PORTRAYER_FINE_QUEUES=$q $B --workload big-soup --samples 64
run "aquarium FINE_QUEUES=$q" PORTRAYER_FINE_QUEUES=$q $B --workload aquarium
run "mirror FINE_QUEUES=$q" PORTRAYER_FINE_QUEUES=$q $B --workload mirror
done > gpurun_out/c67_queues.txt 2>&1
cat gpurun_out/c67_queues.txt
