# round 4, call 40: more fuzz on the final tree (all semantics; larger frames; odd sizes that leave ragged tiles), and the example programs against the reference's committed renders
timeout 1500 python3 tests/fuzz_gpu_parity.py 84000 80 > gpurun_out/c40_fuzz_a.log 2>&1; tail -1 gpurun_out/c40_fuzz_a.log
timeout 1200 python3 tests/fuzz_gpu_parity.py 85000 30 257 131 3 > gpurun_out/c40_fuzz_b.log 2>&1; tail -1 gpurun_out/c40_fuzz_b.log
timeout 1200 python3 tests/fuzz_gpu_parity.py 86000 30 33 17 16 > gpurun_out/c40_fuzz_c.log 2>&1; tail -1 gpurun_out/c40_fuzz_c.log
timeout 1200 bash profiles/examples_vs_goldens.sh > gpurun_out/c40_examples.txt 2>&1; tail -30 gpurun_out/c40_examples.txt
