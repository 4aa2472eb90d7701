python -m pytest tests -m gpu -x -q > gpurun_out/c10_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c10_pytest.log
bash profiles/workloads.sh --no-extras > gpurun_out/c10_workloads.log 2>&1
