# round 5, call 43: the case-split step as the default of every mode: whole suite, every workload, fuzz
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/c43_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c43_pytest.log
grep -n "passed\|failed" gpurun_out/c43_pytest.log | tail -2
timeout 900 bash profiles/workloads.sh --no-extras > gpurun_out/c43_workloads.txt 2>&1; cat gpurun_out/c43_workloads.txt
timeout 1200 python3 tests/fuzz_gpu_parity.py 210000 100 > gpurun_out/c43_fuzz_a.log 2>&1; tail -1 gpurun_out/c43_fuzz_a.log
