# round 4, call 30: the device tree build's search radius (PLOC: each cluster looks R places to either side) on big-soup - walk speed, nodes per ray, build time;
# and the tests of the 1.25 M-triangle scenes at the timed size on their new 5-wave default
run() { name=$1; shift
  env "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']['per_ray']; p=d['config']['prepare_ms']
print('%-34s %9.1f Mray/s %8.3f ms  nodes/ray %.1f tri/ray %.2f  upload+trees %.1f ms  %s' % ('$name', d['value'], d['ms_per_step'], r['inner_nodes'], r['triangle_tests'], p['upload_and_device_trees'], d['roofline']['kernel'][5:58]))"
}
B="python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload big-soup --samples 64"
for R in 16 32 64 128; do run "big-soup x64 PLOC radius $R" PORTRAYER_PLOC_RADIUS=$R $B >> gpurun_out/c30.txt; done
run "big-soup x64 host SAH" PORTRAYER_BUILD=host $B >> gpurun_out/c30.txt
timeout 900 python -m pytest tests/test_gpu_timed_sizes.py -m gpu -q --timeout=600 > gpurun_out/c30_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c30_pytest.log
