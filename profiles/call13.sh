python -m pytest tests/test_shim_replay.py tests/test_examples_extra.py -m gpu -q 2>&1 | tail -15 > gpurun_out/c13_pytest.log
