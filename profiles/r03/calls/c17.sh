# round 3, call 17: the suite on the default build with the chain kernel, then the round's profile set (kernel trace + PMC passes per workload), the default bench line, the workload table
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/c17_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c17_pytest.log
bash profiles/run_profile.sh r03_bigscene --workload big-scene > gpurun_out/c17_prof1.log 2>&1
bash profiles/run_profile.sh r03_hier --workload big-scene --traversal hier > gpurun_out/c17_prof2.log 2>&1
bash profiles/run_profile.sh r03_kd --workload big-scene --traversal kd > gpurun_out/c17_prof3.log 2>&1
bash profiles/run_profile.sh r03_soup64 --workload big-soup --samples 64 > gpurun_out/c17_prof4.log 2>&1
bash profiles/run_profile.sh r03_mirror --workload mirror > gpurun_out/c17_prof5.log 2>&1
bash profiles/run_profile.sh r03_aquarium --workload aquarium > gpurun_out/c17_prof6.log 2>&1
python3 bench.py > gpurun_out/c17_bench.json 2> gpurun_out/c17_bench.err
bash profiles/workloads.sh > gpurun_out/c17_workloads.log 2>&1
