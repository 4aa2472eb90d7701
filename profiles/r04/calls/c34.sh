# round 4, call 34: where the wavefronts' cycles go on the many-triangle workloads (the -DPT_CYCLES build: walks / triangle tests / outside)
bash profiles/cycles.sh "--workload big-soup --samples 64" "--workload big-mesh --samples 64" "--workload cows" "--workload mirror" "--workload big-scene" > gpurun_out/c34_cycles.txt 2>&1
cat gpurun_out/c34_cycles.txt
