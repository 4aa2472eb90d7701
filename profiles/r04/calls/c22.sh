# round 4, call 22: which of this round's three changes to the mesh-free straight-line kernel costs the headline its 2 % (9.63 -> 9.84 ms):
# the walk's watchdog, the bounded chunk-sum loop, the pixel-major chunk sums - each taken away alone, and all three
run() { name=$1; shift
  python3 bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-44s %9.1f Mray/s %8.3f ms (kernel %.3f)' % ('$name', d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"
}
cp portrayer_amd/libportrayer_hip.so /tmp/keep.so
for rep in 1 2; do
for v in keep ab_nw ab_ub ab_cm ab_all3; do
  if [ $v = keep ]; then cp /tmp/keep.so portrayer_amd/libportrayer_hip.so; else cp build/variants/$v/libportrayer_hip.so portrayer_amd/libportrayer_hip.so; fi
  run "$v: big-scene flat" --workload big-scene >> gpurun_out/c22_ab.txt
  run "$v: big-scene hier" --workload big-scene --traversal hier >> gpurun_out/c22_ab.txt
done
done
cp /tmp/keep.so portrayer_amd/libportrayer_hip.so
