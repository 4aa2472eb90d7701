python -m pytest tests -m gpu -x -q > gpurun_out/c90_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c90_pytest.log
bash profiles/workloads.sh --no-extras > gpurun_out/c90_workloads.log 2>&1
