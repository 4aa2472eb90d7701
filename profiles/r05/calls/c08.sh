# round 5, call 8: the tree built through the checked / repaired assembly (the special case of pt_render_kernel.h removed): whole suite, smoke, the two switches that
# reach the repaired instantiation, default bench line
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 -x > gpurun_out/c08_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c08_pytest.log
grep -n "passed\|failed" gpurun_out/c08_pytest.log | tail -2
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/c08_smoke.log 2>&1; tail -1 gpurun_out/c08_smoke.log
for e in PORTRAYER_PARK=0 PORTRAYER_FORK=1; do
  env $e python3 -m pytest tests/test_gpu_render_parity.py tests/test_gpu_textures.py -x -q -m gpu > gpurun_out/c08_$e.txt 2>&1
  echo "$e: $(grep -h 'passed\|failed' gpurun_out/c08_$e.txt | tail -1)"
done
timeout 900 python3 bench.py > gpurun_out/c08_bench.json 2> gpurun_out/c08_bench.err; echo "rc $?" >> gpurun_out/c08_bench.err
python3 -c "
import json
d=json.loads(open('gpurun_out/c08_bench.json').read().strip().split('\n')[-1])
print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['issue_frac'], d['roofline']['f64_frac'], d['roofline']['traffic'], [s['Mray_per_s'] for s in d['secondary']], d['cpu_baseline']['value'])"
