# round 4, call 42: the interpreter's map stage in front of its state machine also in the flat_scene instantiation for plain meshes (mode 1, body in place): parity of every rendering test, speed
python3 -m pytest tests/test_gpu_render_parity.py tests/test_gpu_config_sizes.py tests/test_gpu_textures.py tests/test_gpu_fuzz_slice.py -x -q -m gpu > gpurun_out/c42_tests.txt 2>&1
tail -3 gpurun_out/c42_tests.txt
timeout 900 python3 tests/fuzz_gpu_parity.py 87000 40 > gpurun_out/c42_fuzz.log 2>&1; tail -1 gpurun_out/c42_fuzz.log
bash profiles/variants.sh "" water-glass "water-glass --traversal hier" aquarium cows mirror > gpurun_out/c42_variants.txt 2>&1
cat gpurun_out/c42_variants.txt
