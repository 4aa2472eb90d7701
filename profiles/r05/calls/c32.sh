# round 5, call 32: is it the look-up? the reordering's code without the load of tile_order[] (image order, everything else in place)
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-72s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
B="python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 4 --workload big-scene"
PORTRAYER_ITEM_ORDER=0 $B 2>/dev/null | line "ITEM_ORDER=0"
PORTRAYER_ITEM_ORDER=1 $B 2>/dev/null | line "ITEM_ORDER=1"
PORTRAYER_ITEM_ORDER=1 bash profiles/r05/with_objs.sh "3=build/diag/order_nolookup.o" $B 2>/dev/null | line "ITEM_ORDER=1, no look-up"
