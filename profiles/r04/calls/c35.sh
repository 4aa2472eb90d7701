# round 4, call 35: the three divisions of a triangle test share one refined reciprocal (pt_triangle_hit_e): parity (all GPU tests that render), speed against three plain divisions
python3 -m pytest tests/test_gpu_render_parity.py tests/test_gpu_config_sizes.py tests/test_gpu_fuzz_slice.py tests/test_gpu_device_parity.py -x -q -m gpu > gpurun_out/c35_tests.txt 2>&1
tail -3 gpurun_out/c35_tests.txt
bash profiles/variants.sh "plaindiv" "big-soup --samples 64" "big-mesh --samples 64" "big-soup --samples 64 --traversal hier" big-soup cows mirror "mirror --traversal kd" "cows --traversal kd" > gpurun_out/c35_variants.txt 2>&1
cat gpurun_out/c35_variants.txt
