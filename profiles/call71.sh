python -m pytest tests -m gpu -x -q > gpurun_out/c71_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c71_pytest.log
PORTRAYER_WAVES=4 timeout 1200 python tests/fuzz_gpu_parity.py 130000 400 > gpurun_out/c71_fuzz.log 2>&1
PORTRAYER_WAVES=4 timeout 600 python tests/fuzz_gpu_parity.py 131000 60 64 48 32 >> gpurun_out/c71_fuzz.log 2>&1
PORTRAYER_WAVES=4 timeout 600 python tests/fuzz_gpu_parity.py 132000 40 40 30 64 >> gpurun_out/c71_fuzz.log 2>&1
