# round 4, call 72: the interpreter's switches on the final kernels (fork / join, chunks of a pixel side by side, parked frame)
run() { name=$1; shift
  env "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-44s %9.1f Mray/s %8.3f ms  %s' % ('$name', d['value'], d['ms_per_step'], d['roofline']['kernel'][5:50]))"
}
B="python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1"
for w in aquarium water-glass "aquarium --traversal hier"; do
run "$w default" X=1 $B --workload $w
run "$w FORK=1" PORTRAYER_FORK=1 $B --workload $w
run "$w LANE_CHUNKS=1" PORTRAYER_LANE_CHUNKS=1 $B --workload $w
run "$w LANE_CHUNKS=4" PORTRAYER_LANE_CHUNKS=4 $B --workload $w
run "$w PARK=0" PORTRAYER_PARK=0 $B --workload $w
done > gpurun_out/c72_interp.txt 2>&1
cat gpurun_out/c72_interp.txt
