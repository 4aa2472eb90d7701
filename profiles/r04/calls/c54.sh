# round 4, call 54: exact ties resolved after the test instead of before it (pt_tie_end / pt_tie_before in the wave-uniform leaf tests): parity (suite subset incl. the tie scenes, fuzz in all semantics), speed against the per-test form (tiepertest)
python3 -m pytest tests/test_gpu_render_parity.py tests/test_gpu_config_sizes.py tests/test_gpu_textures.py tests/test_gpu_fuzz_slice.py tests/test_gpu_device_parity.py -x -q -m gpu > gpurun_out/c54_tests.txt 2>&1
grep -n "passed\|failed" gpurun_out/c54_tests.txt | tail -1
timeout 1200 python3 tests/fuzz_gpu_parity.py 93000 60 > gpurun_out/c54_fuzz.log 2>&1; tail -1 gpurun_out/c54_fuzz.log
bash profiles/variants.sh "tiepertest" big-scene big-scene "big-scene --traversal hier" "big-scene --traversal hier" mirror cows "big-soup --samples 64" "mirror --traversal hier" aquarium > gpurun_out/c54_variants.txt 2>&1
cat gpurun_out/c54_variants.txt
