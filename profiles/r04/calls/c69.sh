# round 4, call 69: the k-d semantics handed out through the work queues like the other two (c68): every test with a k-d render, the k-d workloads, the whole suite
python3 -m pytest tests -x -q -m gpu > gpurun_out/c69_pytest.txt 2>&1; grep -h "passed\|failed" gpurun_out/c69_pytest.txt | tail -1
for w in "big-scene" "mirror" "cows" "big-soup" "big-mesh"; do
python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --traversal kd --workload $w 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('--workload %-12s --traversal kd %9.1f Mray/s %8.2f ms/frame' % ('$w', d['value'], d['ms_per_step']))"
done > gpurun_out/c69_kd.txt 2>&1
cat gpurun_out/c69_kd.txt
