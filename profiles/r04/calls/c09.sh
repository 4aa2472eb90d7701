# round 4, call 9: per-ray and per-wavefront counters of the k-d walk (a -DPT_DIAG build): steps, leaves, references, exact tests, culls
cp build/variants/diag/libportrayer_hip.so portrayer_amd/libportrayer_hip.so
for wl in big-scene mirror cows; do
PT_DUMP_COUNTERS=1 timeout 300 python3 bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 0 --workload $wl --traversal kd 2>&1 | grep "^counters" > gpurun_out/c09_counters_$wl.txt
done
