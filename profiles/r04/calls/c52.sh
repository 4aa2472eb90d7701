# round 4, call 52: hierarchical semantics, interpreter kernel - identity levels skipped also when the hit's local ray is rebuilt for its surface (current) against the tree of c51 (prev): parity, speed
FUZZ_MODES=hier timeout 900 python3 tests/fuzz_gpu_parity.py 92000 60 > gpurun_out/c52_fuzz.log 2>&1; tail -1 gpurun_out/c52_fuzz.log
python3 -m pytest tests/test_gpu_render_parity.py tests/test_gpu_config_sizes.py tests/test_gpu_textures.py tests/test_gpu_fuzz_slice.py -x -q -m gpu > gpurun_out/c52_tests.txt 2>&1
grep -n "passed\|failed" gpurun_out/c52_tests.txt | tail -1
bash profiles/variants.sh "prev" "aquarium --traversal hier" "water-glass --traversal hier" "aquarium --traversal hier" "water-glass --traversal hier" "big-scene --traversal hier" "cows --traversal hier" > gpurun_out/c52_variants.txt 2>&1
cat gpurun_out/c52_variants.txt
