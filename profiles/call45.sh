timeout 1500 python tests/fuzz_gpu_parity.py 20000 1500 > gpurun_out/c45_fuzz2.log 2>&1
timeout 900 python tests/fuzz_gpu_parity.py 30000 300 64 48 32 > gpurun_out/c45_fuzz32.log 2>&1
timeout 600 python tests/fuzz_gpu_parity.py 40000 100 40 30 64 > gpurun_out/c45_fuzz64.log 2>&1
PORTRAYER_BUILD=device PORTRAYER_BUILD_MIN=16 timeout 600 python tests/fuzz_gpu_parity.py 50000 150 > gpurun_out/c45_fuzzdev.log 2>&1
PORTRAYER_PARK=0 timeout 600 python tests/fuzz_gpu_parity.py 60000 150 > gpurun_out/c45_fuzzpark0.log 2>&1
PORTRAYER_LDS_STACK=1 timeout 600 python tests/fuzz_gpu_parity.py 70000 100 > gpurun_out/c45_fuzzstack.log 2>&1
