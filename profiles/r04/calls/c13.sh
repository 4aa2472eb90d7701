# round 4, call 13: the multi-rank tests again (both tile-buffer sets are allocated by the first frame), then the whole suite on the in-tree build
timeout 900 python -m pytest tests/test_gpu_multirank.py -m gpu -q --timeout=600 > gpurun_out/c13_pytest_mr.log 2>&1; echo "pytest rc $?" >> gpurun_out/c13_pytest_mr.log
timeout 1800 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/c13_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c13_pytest.log
