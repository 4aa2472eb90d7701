timeout 1500 python tests/fuzz_gpu_parity.py 110000 700 > gpurun_out/c60_fuzz.log 2>&1
timeout 900 python tests/fuzz_gpu_parity.py 111000 120 64 48 32 >> gpurun_out/c60_fuzz.log 2>&1
timeout 900 python tests/fuzz_gpu_parity.py 112000 60 40 30 64 >> gpurun_out/c60_fuzz.log 2>&1
PORTRAYER_BUILD=device PORTRAYER_BUILD_MIN=16 timeout 600 python tests/fuzz_gpu_parity.py 113000 100 >> gpurun_out/c60_fuzz.log 2>&1
PORTRAYER_LDS_STACK=5 timeout 600 python tests/fuzz_gpu_parity.py 114000 100 >> gpurun_out/c60_fuzz.log 2>&1
PORTRAYER_PARK=0 timeout 600 python tests/fuzz_gpu_parity.py 115000 80 >> gpurun_out/c60_fuzz.log 2>&1
