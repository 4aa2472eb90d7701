mkdir -p gpurun_out/c2
python -m pytest tests -m gpu -x -q > gpurun_out/c2/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c2/pytest.log
bash profiles/workloads.sh > gpurun_out/c2/workloads.log 2>&1
bash profiles/diag.sh "--workload big-scene" "--workload big-soup" "--workload mirror --samples 16" "--workload aquarium" "--workload cows" > gpurun_out/c2/diag.log 2>&1
