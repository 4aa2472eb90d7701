python -m pytest tests -m gpu -x -q > gpurun_out/c56_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c56_pytest.log
timeout 1200 python tests/fuzz_gpu_parity.py 100000 400 > gpurun_out/c56_fuzz.log 2>&1
timeout 900 python tests/fuzz_gpu_parity.py 101000 100 64 48 32 >> gpurun_out/c56_fuzz.log 2>&1
timeout 900 python tests/fuzz_gpu_parity.py 102000 60 40 30 64 >> gpurun_out/c56_fuzz.log 2>&1
