# round 4, call 62: the committed tree: whole suite (with the new guard test), smoke, default bench line
timeout 1800 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/c62_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c62_pytest.log
grep -n "passed\|failed" gpurun_out/c62_pytest.log | tail -1
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/c62_smoke.log 2>&1; tail -1 gpurun_out/c62_smoke.log
timeout 900 python3 bench.py > gpurun_out/c62_bench.json 2> gpurun_out/c62_bench.err; echo "rc $?" >> gpurun_out/c62_bench.err
python3 -c "
import json
d=json.loads(open('gpurun_out/c62_bench.json').read().strip().split('\n')[-1])
print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], [s['Mray_per_s'] for s in d['secondary']], d['cpu_baseline']['value'])"
