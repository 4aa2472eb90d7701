# round 4, call 14: the new tests - a slice of the fuzz harness inside the suite, atan2 / acos against glibc itself, the -0 scene on the plain kernels
timeout 900 python -m pytest tests/test_gpu_fuzz_slice.py tests/test_gpu_device_parity.py -m gpu -q --timeout=600 --durations=5 -s > gpurun_out/c14_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c14_pytest.log
timeout 600 python -m pytest tests/test_gpu_render_parity.py -m gpu -q --timeout=600 -k "negative_zero or parallel" >> gpurun_out/c14_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c14_pytest.log
