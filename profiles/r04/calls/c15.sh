# round 4, call 15: (1) what the pop-count watchdog of the wave-uniform walks costs (build/variants/wd against the in-tree build without it),
# (2) the 6-wave hierarchical instantiation of round 3's c48 with that watchdog (limit 200000 pops): does it trip, or does the kernel still not finish
run() { name=$1; shift
  python3 bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-40s %9.1f Mray/s %8.3f ms' % ('$name', d['value'], d['ms_per_step']))"
}
cp portrayer_amd/libportrayer_hip.so /tmp/keep.so
for rep in 1 2; do
cp /tmp/keep.so portrayer_amd/libportrayer_hip.so
run "no watchdog: big-scene flat" --workload big-scene >> gpurun_out/c15_ab.txt
run "no watchdog: big-scene hier" --workload big-scene --traversal hier >> gpurun_out/c15_ab.txt
run "no watchdog: mirror flat" --workload mirror >> gpurun_out/c15_ab.txt
run "no watchdog: cows flat" --workload cows >> gpurun_out/c15_ab.txt
run "no watchdog: big-soup flat" --workload big-soup >> gpurun_out/c15_ab.txt
cp build/variants/wd/libportrayer_hip.so portrayer_amd/libportrayer_hip.so
run "watchdog: big-scene flat" --workload big-scene >> gpurun_out/c15_ab.txt
run "watchdog: big-scene hier" --workload big-scene --traversal hier >> gpurun_out/c15_ab.txt
run "watchdog: mirror flat" --workload mirror >> gpurun_out/c15_ab.txt
run "watchdog: cows flat" --workload cows >> gpurun_out/c15_ab.txt
run "watchdog: big-soup flat" --workload big-soup >> gpurun_out/c15_ab.txt
done
cp build/variants/w6/libportrayer_hip.so portrayer_amd/libportrayer_hip.so
export PORTRAYER_LDS_BUDGET_KB=26
for args in "plain flat 10" "stats hier 10" "plain hier 10" "plain hier 7"; do
  echo "== hang6.py $args" >> gpurun_out/c15_w6.txt
  timeout 60 python3 profiles/r04/hang6.py $args >> gpurun_out/c15_w6.txt 2>&1; echo "rc $?" >> gpurun_out/c15_w6.txt
done
cp /tmp/keep.so portrayer_amd/libportrayer_hip.so
