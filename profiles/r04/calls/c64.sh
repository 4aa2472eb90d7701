# round 4, call 64: a long fuzz run on the committed tree (all semantics, five scene families, counting + plain instantiation), three frame shapes
timeout 2400 python3 tests/fuzz_gpu_parity.py 100000 200 > gpurun_out/c64_fuzz_a.log 2>&1; tail -1 gpurun_out/c64_fuzz_a.log
timeout 1500 python3 tests/fuzz_gpu_parity.py 101000 60 200 120 4 > gpurun_out/c64_fuzz_b.log 2>&1; tail -1 gpurun_out/c64_fuzz_b.log
PORTRAYER_PARK=0 timeout 1500 python3 tests/fuzz_gpu_parity.py 102000 60 > gpurun_out/c64_fuzz_c.log 2>&1; tail -1 gpurun_out/c64_fuzz_c.log
PORTRAYER_WAVES=4 PORTRAYER_KD_WAVES=4 timeout 1500 python3 tests/fuzz_gpu_parity.py 103000 60 > gpurun_out/c64_fuzz_d.log 2>&1; tail -1 gpurun_out/c64_fuzz_d.log
