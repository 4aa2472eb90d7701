# round 4, call 47: a mesh instance entered with three dependent fetches instead of five / six (info as one x4, the mesh record's box inverse and roots in one round trip): parity, speed against the tree before (prev)
python3 -m pytest tests/test_gpu_render_parity.py tests/test_gpu_config_sizes.py tests/test_gpu_textures.py tests/test_gpu_fuzz_slice.py -x -q -m gpu > gpurun_out/c47_tests.txt 2>&1
tail -3 gpurun_out/c47_tests.txt
bash profiles/variants.sh "prev" cows mirror "big-mesh --samples 64" "big-soup --samples 64" "mirror --traversal kd" "cows --traversal kd" "mirror --traversal hier" "cows --traversal hier" aquarium > gpurun_out/c47_variants.txt 2>&1
cat gpurun_out/c47_variants.txt
