# round 3, call 25: the generator's pixel key on the scalar unit where a work item is one pixel
bash profiles/workloads.sh --no-extras > gpurun_out/c25_workloads.log 2>&1
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/c25_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c25_pytest.log
