python -m pytest tests -m gpu -x -q > gpurun_out/c33_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c33_pytest.log
bash profiles/workloads.sh --no-extras > gpurun_out/c33_workloads.log 2>&1
( bash profiles/pmc_quick.sh "WRITE_SIZE" --no-extras --workload big-scene; bash profiles/pmc_quick.sh "FETCH_SIZE" --no-extras --workload big-scene ) > gpurun_out/c33_pmc.log 2>&1
timeout 900 python tests/fuzz_gpu_parity.py 11000 60 > gpurun_out/c33_fuzz.log 2>&1
timeout 600 python tests/fuzz_gpu_parity.py 12000 30 48 36 32 >> gpurun_out/c33_fuzz.log 2>&1
