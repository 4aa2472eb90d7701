# round 5, call 58: the final tree: whole suite, smoke, the driver's bench command, every workload, fuzz (default switches, PARK=0), profile sets of the mirror and the dielectric kernels
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/c58_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c58_pytest.log
grep -n "passed\|failed" gpurun_out/c58_pytest.log | tail -2
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/c58_smoke.log 2>&1; tail -1 gpurun_out/c58_smoke.log
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/c58_bench.json 2> gpurun_out/c58_bench.err; echo "rc $?" >> gpurun_out/c58_bench.err
python3 -c "
import json
d=json.loads(open('gpurun_out/c58_bench.json').read().strip().split('\n')[-1])
r=d['roofline']
print('bench', d['value'], d['ms_per_step'], 'frac', r['frac'], 'issue', r['issue_frac'], 'f64', r['f64_frac'], 'traffic', r['traffic'], [s['Mray_per_s'] for s in d['secondary']], d['cpu_baseline']['value'])"
timeout 900 bash profiles/workloads.sh --no-extras > gpurun_out/c58_workloads.txt 2>&1; cat gpurun_out/c58_workloads.txt
timeout 1200 python3 tests/fuzz_gpu_parity.py 230000 120 > gpurun_out/c58_fuzz_a.log 2>&1; tail -1 gpurun_out/c58_fuzz_a.log
PORTRAYER_PARK=0 timeout 900 python3 tests/fuzz_gpu_parity.py 231000 40 > gpurun_out/c58_fuzz_b.log 2>&1; tail -1 gpurun_out/c58_fuzz_b.log
timeout 1500 bash profiles/run_profile.sh r05b_mirror --workload mirror > /dev/null 2>&1
timeout 1500 bash profiles/run_profile.sh r05b_aquarium --workload aquarium > /dev/null 2>&1
