# round 5, call 78: the last tree of round 5 (after the k-d general-case block): whole suite, smoke, the bench command three times, workloads, fuzz
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/c78_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c78_pytest.log
grep -n "passed\|failed" gpurun_out/c78_pytest.log | tail -2
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/c78_smoke.log 2>&1; tail -1 gpurun_out/c78_smoke.log
for k in 1 2 3; do
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/c78_bench_$k.json 2> gpurun_out/c78_bench.err
python3 -c "
import json
d=json.loads(open('gpurun_out/c78_bench_$k.json').read().strip().split('\n')[-1])
r=d['roofline']
print('bench', d['value'], d['ms_per_step'], 'frac', r['frac'], 'issue', r['issue_frac'], 'f64', r['f64_frac'], 'traffic', r['traffic'], [s['Mray_per_s'] for s in d['secondary']], d['cpu_baseline']['value'])"
done
timeout 900 bash profiles/workloads.sh --no-extras > gpurun_out/c78_workloads.txt 2>&1; cat gpurun_out/c78_workloads.txt
timeout 900 python3 tests/fuzz_gpu_parity.py 299000 100 > gpurun_out/c78_fuzz.log 2>&1; tail -1 gpurun_out/c78_fuzz.log
