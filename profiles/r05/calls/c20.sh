# round 5, call 20: the tree with the k-d walk's top-level bounds recomputed (no HBM columns): whole suite, k-d timings, the k-d profile set
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 -x > gpurun_out/c20_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c20_pytest.log
grep -n "passed\|failed" gpurun_out/c20_pytest.log | tail -2
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-58s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
for a in "--workload big-scene --traversal kd" "--workload mirror --traversal kd" "--workload cows --traversal kd" "--workload big-soup --traversal kd"; do
  python3 bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 $a 2>/dev/null | line "$a"
done > gpurun_out/c20_kd.txt 2>&1; cat gpurun_out/c20_kd.txt
timeout 1500 bash profiles/run_profile.sh r05_kd --workload big-scene --traversal kd > /dev/null 2>&1
python3 profiles/summarise.py gpurun_out/prof_r05_kd > gpurun_out/r05_kd_pmc.json 2> gpurun_out/c20_summarise.err; head -c 600 gpurun_out/r05_kd_pmc.json; tail -2 gpurun_out/c20_summarise.err
