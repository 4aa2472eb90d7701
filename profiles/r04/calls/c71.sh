# round 4, call 71: the very long launch (3840x2160x256, C5) - guided batches (the policy for > 2048 items per resident wavefront) against the queues, on the final kernels
run() { name=$1; shift
  env "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-48s %9.1f Mray/s %9.3f ms' % ('$name', d['value'], d['ms_per_step']))"
}
B="python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 --workload big-scene --width 3840 --height 2160 --samples 256"
run "C5 flat default" X=1 $B
run "C5 flat FINE_QUEUES=16" PORTRAYER_FINE_QUEUES=16 $B
run "C5 flat FINE_QUEUES=0" PORTRAYER_FINE_QUEUES=0 $B
run "C5 hier default" X=1 $B --traversal hier
run "C5 hier FINE_QUEUES=16" PORTRAYER_FINE_QUEUES=16 $B --traversal hier
run "C5 kd default" X=1 $B --traversal kd
run "C5 kd FINE_QUEUES=0" PORTRAYER_FINE_QUEUES=0 $B --traversal kd
