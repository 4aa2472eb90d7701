bash profiles/cycles.sh "--workload big-scene" "--workload big-scene --traversal hier" "--workload mirror" "--workload big-soup --samples 64" "--workload aquarium" > gpurun_out/c17_cycles.log 2>&1
bash profiles/diag.sh "--workload big-scene" "--workload big-soup --samples 64" "--workload mirror" "--workload aquarium" "--workload big-scene --traversal kd" > gpurun_out/c17_diag.log 2>&1
