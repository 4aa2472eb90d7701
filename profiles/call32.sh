python -m pytest tests -m gpu -x -q > gpurun_out/c32_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c32_pytest.log
bash profiles/workloads.sh --no-extras > gpurun_out/c32_workloads.log 2>&1
bash profiles/run_profile.sh r02_bigscene --workload big-scene > /dev/null 2>&1
bash profiles/run_profile.sh r02_hier --workload big-scene --traversal hier > /dev/null 2>&1
bash profiles/run_profile.sh r02_kd --workload big-scene --traversal kd > /dev/null 2>&1
bash profiles/run_profile.sh r02_soup64 --workload big-soup --samples 64 > /dev/null 2>&1
bash profiles/run_profile.sh r02_mirror --workload mirror > /dev/null 2>&1
bash profiles/run_profile.sh r02_aquarium --workload aquarium > /dev/null 2>&1
for t in r02_bigscene r02_hier r02_kd r02_soup64 r02_mirror r02_aquarium; do python3 profiles/digest.py $t; done > gpurun_out/c32_digest.log 2>&1
bash profiles/diag.sh "--workload big-scene" "--workload big-soup --samples 64" "--workload mirror" "--workload aquarium" "--workload cows" "--workload big-scene --traversal kd" > gpurun_out/c32_diag.log 2>&1
bash profiles/timeline.sh "--workload big-scene" "--workload big-soup --samples 64" "--workload mirror" "--workload aquarium" "--workload cows" "--workload big-scene --traversal kd" > gpurun_out/c32_timeline.log 2>&1
bash profiles/cycles.sh "--workload big-scene" "--workload aquarium" "--workload mirror" "--workload big-soup --samples 64" > gpurun_out/c32_cycles.log 2>&1
