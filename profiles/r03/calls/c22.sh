# round 3, call 22: the argument block re-read per work item (scalar loads instead of scalar registers spilled into vector lanes)
bash profiles/workloads.sh --no-extras > gpurun_out/c22_workloads.log 2>&1
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/c22_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c22_pytest.log
