# round 4, call 43: which kernel water-glass runs, and the map routine inline in every kernel of every mode (m1inl) against the shipped tree
run() { name=$1; shift
  env "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-44s %9.1f Mray/s %8.3f ms  %s' % ('$name', d['value'], d['ms_per_step'], d['roofline']['kernel'][5:70]))"
}
B="python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1"
for lib in current m1inl; do
  if [ $lib != current ]; then cp portrayer_amd/libportrayer_hip.so /tmp/keep.so; cp build/variants/$lib/libportrayer_hip.so portrayer_amd/libportrayer_hip.so; fi
  run "$lib water-glass" X=1 $B --workload water-glass
  run "$lib water-glass" X=1 $B --workload water-glass
  run "$lib water-glass hier" X=1 $B --workload water-glass --traversal hier
  run "$lib aquarium" X=1 $B --workload aquarium
  run "$lib aquarium hier" X=1 $B --workload aquarium --traversal hier
  if [ $lib != current ]; then cp /tmp/keep.so portrayer_amd/libportrayer_hip.so; fi
done > gpurun_out/c43.txt 2>&1
cat gpurun_out/c43.txt
