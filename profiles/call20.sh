python -m pytest tests -m gpu -x -q > gpurun_out/c20_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c20_pytest.log
timeout 1500 python tests/fuzz_gpu_parity.py 7000 250 > gpurun_out/c20_fuzz2.log 2>&1
timeout 900 python tests/fuzz_gpu_parity.py 8000 80 64 48 32 > gpurun_out/c20_fuzz32.log 2>&1
timeout 600 python tests/fuzz_gpu_parity.py 9000 40 40 30 64 > gpurun_out/c20_fuzz64.log 2>&1
