# round 3, call 23: item geometry by shifts and scalar multiply-high divisions; the argument block re-read per interpreter pass
bash profiles/workloads.sh --no-extras > gpurun_out/c23_workloads.log 2>&1
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/c23_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c23_pytest.log
