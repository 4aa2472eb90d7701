# round 4, call 49: the next item's ticket drawn while the current item is rendered (straight-line kernels, fine queues) against drawing it when needed (noticket): parity, speed
python3 -m pytest tests/test_gpu_render_parity.py tests/test_gpu_config_sizes.py tests/test_gpu_fuzz_slice.py tests/test_gpu_multirank.py tests/test_gpu_timed_sizes.py -x -q -m gpu > gpurun_out/c49_tests.txt 2>&1
tail -3 gpurun_out/c49_tests.txt
bash profiles/variants.sh "noticket" big-scene big-scene "big-scene --traversal hier" mirror cows "big-soup --samples 64" "big-scene --share 8" > gpurun_out/c49_variants.txt 2>&1
cat gpurun_out/c49_variants.txt
