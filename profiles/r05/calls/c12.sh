# round 5, call 12: two frames in flight on two streams (ABI 8): the tests, then what it buys - one share of an 8-way split frame by frame against two at a time,
# the full frame likewise, and 8 ranks sharing this GPU through pt_node with and without the second stream
timeout 900 python3 -m pytest tests/test_gpu_multirank.py -x -q -m gpu > gpurun_out/c12_tests.txt 2>&1; grep -h "passed\|failed" gpurun_out/c12_tests.txt | tail -1
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-58s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
for a in "--workload big-scene --share 8 --share-rank 0" "--workload big-scene --share 8 --share-rank 0 --overlap" "--workload big-scene --share 8 --share-rank 5" "--workload big-scene --share 8 --share-rank 5 --overlap" "--workload big-scene" "--workload big-scene --overlap" "--workload mirror" "--workload mirror --overlap" "--workload cows" "--workload cows --overlap"; do
  python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 $a 2>/dev/null | line "$a"
done > gpurun_out/c12_overlap.txt 2>&1
for e in 0 1; do
  PORTRAYER_NODE_ONE_STREAM=$e python3 bench.py --gpus 8 --same-device --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | line "--gpus 8 --same-device PORTRAYER_NODE_ONE_STREAM=$e"
done >> gpurun_out/c12_overlap.txt 2>&1
cat gpurun_out/c12_overlap.txt
