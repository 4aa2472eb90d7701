# round 4, call 32: the octant-sorted slab test inside mesh instances (sc.mesh_oct): parity with it forced on for every scene, speed on the many-triangle workloads
PORTRAYER_MESH_OCT=1 python3 -m pytest tests/test_gpu_render_parity.py tests/test_gpu_config_sizes.py tests/test_gpu_fuzz_slice.py -x -q -m gpu > gpurun_out/c32_tests.txt 2>&1
tail -3 gpurun_out/c32_tests.txt
run() { name=$1; shift
  env "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-44s %9.1f Mray/s %8.3f ms  %s' % ('$name', d['value'], d['ms_per_step'], d['roofline']['kernel'][5:60]))"
}
B="python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1"
for o in 0 1; do
run "big-soup x64 MESH_OCT=$o" PORTRAYER_MESH_OCT=$o $B --workload big-soup --samples 64
run "big-mesh x64 MESH_OCT=$o" PORTRAYER_MESH_OCT=$o $B --workload big-mesh --samples 64
run "big-soup x64 hier MESH_OCT=$o" PORTRAYER_MESH_OCT=$o $B --workload big-soup --samples 64 --traversal hier
run "big-soup x16 MESH_OCT=$o" PORTRAYER_MESH_OCT=$o $B --workload big-soup
run "cows MESH_OCT=$o" PORTRAYER_MESH_OCT=$o $B --workload cows
run "mirror MESH_OCT=$o" PORTRAYER_MESH_OCT=$o $B --workload mirror
done > gpurun_out/c32_oct.txt 2>&1
cat gpurun_out/c32_oct.txt
