# round 4, call 28: big-soup (1.25 M triangles) at 1920x1080x64: the device-built tree (PLOC) against the host's binned-SAH tree, 3 / 4 waves, and the k-d semantics at 4 waves for its traffic
run() { name=$1; shift
  env "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']['per_ray']
print('%-44s %9.1f Mray/s %8.3f ms  nodes/ray %.1f tri/ray %.2f  %s' % ('$name', d['value'], d['ms_per_step'], r['inner_nodes'], r['triangle_tests'], d['roofline']['kernel'][5:60]))"
}
B="python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1"
run "big-soup x64 device tree" X=1 $B --workload big-soup --samples 64 >> gpurun_out/c28.txt
run "big-soup x64 host SAH tree" PORTRAYER_BUILD=host $B --workload big-soup --samples 64 >> gpurun_out/c28.txt
run "big-soup x64 device tree, 3 waves" PORTRAYER_WAVES=3 $B --workload big-soup --samples 64 >> gpurun_out/c28.txt
run "big-mesh x64" X=1 $B --workload big-mesh --samples 64 >> gpurun_out/c28.txt
run "big-scene kd 4 waves" PORTRAYER_KD_WAVES=4 $B --workload big-scene --traversal kd >> gpurun_out/c28.txt
PORTRAYER_KD_WAVES=4 bash profiles/pmc_quick.sh "FETCH_SIZE" --no-extras --workload big-scene --traversal kd > gpurun_out/c28_kd4_fetch.txt 2>&1
PORTRAYER_KD_WAVES=4 bash profiles/pmc_quick.sh "WRITE_SIZE" --no-extras --workload big-scene --traversal kd > gpurun_out/c28_kd4_write.txt 2>&1
