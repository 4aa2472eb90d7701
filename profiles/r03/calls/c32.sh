# round 3, call 32: no division by an attenuation of exactly 1.0 (lights without falloff)
bash profiles/workloads.sh --no-extras > gpurun_out/c32_workloads.log 2>&1
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/c32_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c32_pytest.log
timeout 600 python3 tests/fuzz_gpu_parity.py 21000 60 > gpurun_out/c32_fuzz.log 2>&1
