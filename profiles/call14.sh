python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/c14_pytest.log
bash profiles/workloads.sh --no-extras > gpurun_out/c14_workloads.log 2>&1
