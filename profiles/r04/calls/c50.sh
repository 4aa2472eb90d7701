# round 4, call 50: hierarchical semantics: a node's own level's inverse fetched with its path record (one round trip per leaf test instead of two): parity, speed against the tree before (noown)
python3 -m pytest tests/test_gpu_render_parity.py tests/test_gpu_config_sizes.py tests/test_gpu_textures.py tests/test_gpu_fuzz_slice.py -x -q -m gpu > gpurun_out/c50_tests.txt 2>&1
grep -n "passed\|failed" gpurun_out/c50_tests.txt | tail -1
FUZZ_MODES=hier timeout 900 python3 tests/fuzz_gpu_parity.py 90000 60 > gpurun_out/c50_fuzz.log 2>&1; tail -1 gpurun_out/c50_fuzz.log
bash profiles/variants.sh "noown" "big-scene --traversal hier" "big-scene --traversal hier" "mirror --traversal hier" "cows --traversal hier" "aquarium --traversal hier" "water-glass --traversal hier" "big-soup --samples 64 --traversal hier" > gpurun_out/c50_variants.txt 2>&1
cat gpurun_out/c50_variants.txt
