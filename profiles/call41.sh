PT_DUMP_COUNTERS=1 python3 bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 0 --workload big-scene --traversal kd 2>&1 | grep "^counters" > gpurun_out/c41_kd_counters.log
PT_DUMP_COUNTERS=1 python3 bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 0 --workload big-scene 2>&1 | grep "^counters" >> gpurun_out/c41_kd_counters.log
bash profiles/cycles.sh "--workload big-scene --traversal kd" >> gpurun_out/c41_kd_counters.log 2>&1
