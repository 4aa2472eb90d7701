# round 5, call 25: the final tree: whole suite, smoke, the driver's bench command, every workload, fuzz (default switches, PARK=0, two streams forced), the headline's profile set
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/c25_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c25_pytest.log
grep -n "passed\|failed" gpurun_out/c25_pytest.log | tail -2
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/c25_smoke.log 2>&1; tail -1 gpurun_out/c25_smoke.log
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/c25_bench.json 2> gpurun_out/c25_bench.err; echo "rc $?" >> gpurun_out/c25_bench.err
python3 -c "
import json
d=json.loads(open('gpurun_out/c25_bench.json').read().strip().split('\n')[-1])
r=d['roofline']
print('bench', d['value'], d['ms_per_step'], 'frac', r['frac'], 'issue', r['issue_frac'], 'f64', r['f64_frac'], 'traffic', r['traffic'], 'two in flight', d['config'].get('two_frames_in_flight'), [s['Mray_per_s'] for s in d['secondary']], d['cpu_baseline']['value'])"
timeout 900 bash profiles/workloads.sh --no-extras > gpurun_out/c25_workloads.txt 2>&1; cat gpurun_out/c25_workloads.txt
timeout 1200 python3 tests/fuzz_gpu_parity.py 200000 120 > gpurun_out/c25_fuzz_a.log 2>&1; tail -1 gpurun_out/c25_fuzz_a.log
PORTRAYER_PARK=0 timeout 900 python3 tests/fuzz_gpu_parity.py 201000 40 > gpurun_out/c25_fuzz_b.log 2>&1; tail -1 gpurun_out/c25_fuzz_b.log
timeout 1500 bash profiles/run_profile.sh r05_bigscene --workload big-scene > /dev/null 2>&1
ls gpurun_out/prof_r05_bigscene/*/ | head
