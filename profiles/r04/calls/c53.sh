# round 4, call 53: the committed tree: whole suite, smoke, default bench line
timeout 1800 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/c53_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c53_pytest.log
grep -n "passed\|failed" gpurun_out/c53_pytest.log | tail -1
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/c53_smoke.log 2>&1; tail -1 gpurun_out/c53_smoke.log
timeout 900 python3 bench.py > gpurun_out/c53_bench.json 2> gpurun_out/c53_bench.err; echo "rc $?" >> gpurun_out/c53_bench.err
tail -c 300 gpurun_out/c53_bench.json
