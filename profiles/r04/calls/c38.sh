# round 4, call 38: fuzz parity on the final tree (maps before the state machine, octant-sorted mesh steps, edge-form triangles): all three semantics,
# five scene families, counting and plain instantiation each; 2, 13 and 64 samples per pixel
timeout 1500 python3 tests/fuzz_gpu_parity.py 80000 60 > gpurun_out/c38_fuzz_a.log 2>&1; tail -2 gpurun_out/c38_fuzz_a.log
timeout 900 python3 tests/fuzz_gpu_parity.py 81000 20 96 64 13 > gpurun_out/c38_fuzz_b.log 2>&1; tail -2 gpurun_out/c38_fuzz_b.log
timeout 900 python3 tests/fuzz_gpu_parity.py 82000 12 64 48 64 > gpurun_out/c38_fuzz_c.log 2>&1; tail -2 gpurun_out/c38_fuzz_c.log
