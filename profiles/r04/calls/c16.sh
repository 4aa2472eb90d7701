# round 4, call 16: (1) the lighter watchdog (mesh-free walk: outer loop only) against the in-tree build without it; (2) HEAD's 6-wave build (no watchdog in the
# flat / hierarchical walks) on the scene that never finished in round 3: does it still hang
run() { name=$1; shift
  python3 bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-40s %9.1f Mray/s %8.3f ms' % ('$name', d['value'], d['ms_per_step']))"
}
cp portrayer_amd/libportrayer_hip.so /tmp/keep.so
for rep in 1 2; do
cp /tmp/keep.so portrayer_amd/libportrayer_hip.so
run "no watchdog: big-scene flat" --workload big-scene >> gpurun_out/c16_ab.txt
run "no watchdog: big-scene hier" --workload big-scene --traversal hier >> gpurun_out/c16_ab.txt
cp build/variants/wd3/libportrayer_hip.so portrayer_amd/libportrayer_hip.so
run "outer-loop watchdog: big-scene flat" --workload big-scene >> gpurun_out/c16_ab.txt
run "outer-loop watchdog: big-scene hier" --workload big-scene --traversal hier >> gpurun_out/c16_ab.txt
done
cp build/variants/w6nw/libportrayer_hip.so portrayer_amd/libportrayer_hip.so
export PORTRAYER_LDS_BUDGET_KB=26
for args in "plain flat 10" "stats hier 10" "plain hier 10" "plain hier 7"; do
  echo "== (6 waves, no watchdog) hang6.py $args" >> gpurun_out/c16_w6.txt
  timeout 40 python3 profiles/r04/hang6.py $args >> gpurun_out/c16_w6.txt 2>&1; echo "rc $?" >> gpurun_out/c16_w6.txt
done
cp /tmp/keep.so portrayer_amd/libportrayer_hip.so
cp build/variants/w6/libportrayer_hip.so portrayer_amd/libportrayer_hip.so
for args in "plain hier 10" "plain hier 7" "plain flat 10"; do
  echo "== (6 waves, outer-loop watchdog, limit 200000) hang6.py $args" >> gpurun_out/c16_w6.txt
  timeout 40 python3 profiles/r04/hang6.py $args >> gpurun_out/c16_w6.txt 2>&1; echo "rc $?" >> gpurun_out/c16_w6.txt
done
cp /tmp/keep.so portrayer_amd/libportrayer_hip.so
