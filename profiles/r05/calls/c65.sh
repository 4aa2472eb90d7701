# round 5, call 65: the k-d split's common case first; one instantiation below k-d leaves: whole suite, smoke, bench, workloads (+ k-d ones), fuzz
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/c65_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c65_pytest.log
grep -n "passed\|failed" gpurun_out/c65_pytest.log | tail -2
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/c65_smoke.log 2>&1; tail -1 gpurun_out/c65_smoke.log
for k in 1 2 3; do
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/c65_bench_$k.json 2> gpurun_out/c65_bench.err
python3 -c "
import json
d=json.loads(open('gpurun_out/c65_bench_$k.json').read().strip().split('\n')[-1])
r=d['roofline']
print('bench', d['value'], d['ms_per_step'], 'frac', r['frac'], 'issue', r['issue_frac'], 'f64', r['f64_frac'], 'traffic', r['traffic'], [s['Mray_per_s'] for s in d['secondary']], d['cpu_baseline']['value'])"
done
timeout 900 bash profiles/workloads.sh --no-extras > gpurun_out/c65_workloads.txt 2>&1; cat gpurun_out/c65_workloads.txt
timeout 900 python3 tests/fuzz_gpu_parity.py 260000 120 > gpurun_out/c65_fuzz.log 2>&1; tail -1 gpurun_out/c65_fuzz.log
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-72s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
for a in "--workload big-scene --traversal kd" "--workload mirror --traversal kd" "--workload cows --traversal kd" "--workload big-soup --traversal kd" "--workload aquarium --traversal kd"; do
  python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "$a"
done > gpurun_out/c65_kd.txt 2>&1
cat gpurun_out/c65_kd.txt
timeout 1500 bash profiles/run_profile.sh r05b_kd --workload big-scene --traversal kd > /dev/null 2>&1
