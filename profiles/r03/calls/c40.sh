# round 3, call 40: long fuzz runs on the final build: 2 samples (32 pixels a wavefront), 8 samples (8 pixels), 64 samples (one pixel)
timeout 1200 python3 tests/fuzz_gpu_parity.py 30000 400 > gpurun_out/c40_fuzz_x2.log 2>&1
timeout 900 python3 tests/fuzz_gpu_parity.py 31000 120 96 64 8 > gpurun_out/c40_fuzz_x8.log 2>&1
timeout 900 python3 tests/fuzz_gpu_parity.py 32000 40 64 48 64 > gpurun_out/c40_fuzz_x64.log 2>&1
