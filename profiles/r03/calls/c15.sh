# round 3, call 15: the round's profile set on the final kernels (kernel trace + PMC passes per workload), cycle split, the default bench line
bash profiles/run_profile.sh r03_bigscene --workload big-scene > gpurun_out/c15_prof1.log 2>&1
bash profiles/run_profile.sh r03_hier --workload big-scene --traversal hier > gpurun_out/c15_prof2.log 2>&1
bash profiles/run_profile.sh r03_kd --workload big-scene --traversal kd > gpurun_out/c15_prof3.log 2>&1
bash profiles/run_profile.sh r03_soup64 --workload big-soup --samples 64 > gpurun_out/c15_prof4.log 2>&1
bash profiles/run_profile.sh r03_mirror --workload mirror > gpurun_out/c15_prof5.log 2>&1
bash profiles/run_profile.sh r03_aquarium --workload aquarium > gpurun_out/c15_prof6.log 2>&1
bash profiles/cycles.sh "--workload big-scene" "--workload big-scene --traversal hier" "--workload big-soup --samples 64" "--workload mirror" "--workload aquarium" > gpurun_out/c15_cycles.log 2>&1
python3 bench.py > gpurun_out/c15_bench.json 2> gpurun_out/c15_bench.err
bash profiles/workloads.sh --no-extras > gpurun_out/c15_workloads.log 2>&1
