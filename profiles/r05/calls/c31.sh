# round 5, call 31: the hand-out order by cost CLASS (stable: image order inside a class) at three coarsenesses, against the full sort and against image order
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-72s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
for cfg in "PORTRAYER_ITEM_ORDER=0" "PORTRAYER_ITEM_ORDER=1 PORTRAYER_ITEM_ORDER_SHIFT=-1" "PORTRAYER_ITEM_ORDER=1 PORTRAYER_ITEM_ORDER_SHIFT=0" "PORTRAYER_ITEM_ORDER=1 PORTRAYER_ITEM_ORDER_SHIFT=1" "PORTRAYER_ITEM_ORDER=1 PORTRAYER_ITEM_ORDER_SHIFT=2"; do
for a in "--workload big-scene --share 8 --share-rank 0" "--workload big-scene"; do
  env $cfg python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 4 $a 2>/dev/null | line "$cfg $a"
done; done > gpurun_out/c31_item_order_classes.txt 2>&1
cat gpurun_out/c31_item_order_classes.txt
