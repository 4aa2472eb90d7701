# round 5, call 35: the hand-out order by groups of tiles through the scalar cache: parity test, then image order against cost classes / full sort
timeout 1200 python3 -m pytest tests/test_gpu_render_parity.py -k "dearest" tests/test_gpu_multirank.py -q -m gpu > gpurun_out/c35_tests.txt 2>&1; grep -h "passed\|failed" gpurun_out/c35_tests.txt | tail -1
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-72s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
for cfg in "PORTRAYER_ITEM_ORDER=0" "PORTRAYER_ITEM_ORDER=1 PORTRAYER_ITEM_ORDER_SHIFT=-1" "PORTRAYER_ITEM_ORDER=1 PORTRAYER_ITEM_ORDER_SHIFT=0" "PORTRAYER_ITEM_ORDER=1 PORTRAYER_ITEM_ORDER_SHIFT=1"; do
for a in "--workload big-scene --share 8 --share-rank 0" "--workload big-scene --share 8 --share-rank 5" "--workload big-scene" "--workload mirror" "--workload cows"; do
  env $cfg python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 4 $a 2>/dev/null | line "$cfg $a"
done; done > gpurun_out/c35_item_order_groups.txt 2>&1
cat gpurun_out/c35_item_order_groups.txt
