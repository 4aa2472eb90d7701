# round 4, call 44: the map stage in front of the state machine in the mesh-free flat_scene instantiation (mode 3, body in place): parity, speed
python3 -m pytest tests/test_gpu_render_parity.py tests/test_gpu_config_sizes.py tests/test_gpu_textures.py tests/test_gpu_fuzz_slice.py -x -q -m gpu > gpurun_out/c44_tests.txt 2>&1
tail -3 gpurun_out/c44_tests.txt
timeout 900 python3 tests/fuzz_gpu_parity.py 88000 40 > gpurun_out/c44_fuzz.log 2>&1; tail -1 gpurun_out/c44_fuzz.log
bash profiles/variants.sh "" water-glass water-glass "water-glass --traversal hier" aquarium big-scene > gpurun_out/c44_variants.txt 2>&1
cat gpurun_out/c44_variants.txt
