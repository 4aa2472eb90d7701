# round 5, call 30: frames of a sequence handed out dearest tiles first: the test, the multi-rank tests, then what it buys (one share, the full frame, 8 ranks on one GPU)
timeout 1200 python3 -m pytest tests/test_gpu_render_parity.py -k "dearest or example_matches" tests/test_gpu_multirank.py -q -m gpu -x > gpurun_out/c30_tests.txt 2>&1; grep -h "passed\|failed" gpurun_out/c30_tests.txt | tail -1
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-66s %9.1f Mray/s %8.3f ms/frame kernel %8.3f' % ('$1', d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"; }
for o in 0 1; do
for a in "--workload big-scene --share 8 --share-rank 0" "--workload big-scene --share 8 --share-rank 0 --overlap" "--workload big-scene --share 8 --share-rank 5" "--workload big-scene" "--workload big-scene --traversal kd" "--workload mirror" "--workload cows" "--workload aquarium"; do
  PORTRAYER_ITEM_ORDER=$o python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 4 $a 2>/dev/null | line "ITEM_ORDER=$o $a"
done; done > gpurun_out/c30_item_order.txt 2>&1
for o in 0 1; do PORTRAYER_ITEM_ORDER=$o python3 bench.py --gpus 8 --same-device --steps 10 --warmup 4 --no-cpu-baseline 2>/dev/null | line "ITEM_ORDER=$o --gpus 8 --same-device"; done >> gpurun_out/c30_item_order.txt 2>&1
cat gpurun_out/c30_item_order.txt
