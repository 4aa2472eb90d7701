# round 4, call 25: 6 waves per SIMD for the mesh-free straight-line kernels once more, now that the hang of round 3's 6-wave build is understood
# (80 registers, 26 KB of LDS a block): big-scene in the flat_scene and hierarchical semantics, the 4K configuration, and the parity tests of those kernels
run() { name=$1; shift
  env "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-44s %9.1f Mray/s %8.3f ms (kernel %.3f)' % ('$name', d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"
}
B="python3 bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2"
cp portrayer_amd/libportrayer_hip.so /tmp/keep.so
for rep in 1 2; do
cp /tmp/keep.so portrayer_amd/libportrayer_hip.so
run "5 waves: flat" X=1 $B --workload big-scene >> gpurun_out/c25_ab.txt
run "5 waves: hier" X=1 $B --workload big-scene --traversal hier >> gpurun_out/c25_ab.txt
cp build/variants/w6/libportrayer_hip.so portrayer_amd/libportrayer_hip.so
run "6 waves (26 KB LDS): flat" PORTRAYER_LDS_BUDGET_KB=26 $B --workload big-scene >> gpurun_out/c25_ab.txt
run "6 waves (26 KB LDS): hier" PORTRAYER_LDS_BUDGET_KB=26 $B --workload big-scene --traversal hier >> gpurun_out/c25_ab.txt
done
run "6 waves: flat 3840x2160x256" PORTRAYER_LDS_BUDGET_KB=26 python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 --workload big-scene --width 3840 --height 2160 --samples 256 >> gpurun_out/c25_ab.txt
cp /tmp/keep.so portrayer_amd/libportrayer_hip.so
run "5 waves: flat 3840x2160x256" X=1 python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 --workload big-scene --width 3840 --height 2160 --samples 256 >> gpurun_out/c25_ab.txt
