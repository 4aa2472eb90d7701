# round 3, call 36: the colour so far and the light's distance wait in LDS while the shadow ray is walked
bash profiles/workloads.sh --no-extras > gpurun_out/c36_workloads.log 2>&1
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/c36_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c36_pytest.log
timeout 600 python3 tests/fuzz_gpu_parity.py 24000 60 > gpurun_out/c36_fuzz.log 2>&1
