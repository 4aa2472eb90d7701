timeout 1700 python tests/fuzz_gpu_parity.py 200000 1500 > gpurun_out/c81_fuzz_a.log 2>&1
PORTRAYER_WAVES=4 timeout 900 python tests/fuzz_gpu_parity.py 210000 500 > gpurun_out/c81_fuzz_b.log 2>&1
PORTRAYER_FINE_QUEUES=0 timeout 900 python tests/fuzz_gpu_parity.py 220000 400 > gpurun_out/c81_fuzz_c.log 2>&1
PORTRAYER_BUILD=device PORTRAYER_BUILD_MIN=16 PORTRAYER_WAVES=4 timeout 900 python tests/fuzz_gpu_parity.py 230000 300 > gpurun_out/c81_fuzz_d.log 2>&1
timeout 900 python tests/fuzz_gpu_parity.py 240000 300 64 48 32 > gpurun_out/c81_fuzz_e.log 2>&1
timeout 900 python tests/fuzz_gpu_parity.py 250000 150 40 30 64 > gpurun_out/c81_fuzz_f.log 2>&1
