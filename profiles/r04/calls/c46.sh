# round 4, call 46: triangle records in leaf order (tri_leaf: no look-up through bvh_items in the wave-uniform walks): parity, speed against the look-up (viaitems)
python3 -m pytest tests/test_gpu_render_parity.py tests/test_gpu_config_sizes.py tests/test_gpu_textures.py tests/test_gpu_fuzz_slice.py tests/test_gpu_timed_sizes.py -x -q -m gpu > gpurun_out/c46_tests.txt 2>&1
tail -3 gpurun_out/c46_tests.txt
timeout 900 python3 tests/fuzz_gpu_parity.py 89000 40 > gpurun_out/c46_fuzz.log 2>&1; tail -1 gpurun_out/c46_fuzz.log
bash profiles/variants.sh "viaitems" "big-soup --samples 64" "big-mesh --samples 64" "big-soup --samples 64 --traversal hier" big-soup cows mirror "mirror --traversal kd" "cows --traversal kd" > gpurun_out/c46_variants.txt 2>&1
cat gpurun_out/c46_variants.txt
