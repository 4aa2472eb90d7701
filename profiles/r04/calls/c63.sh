# round 4, call 63: the hierarchical profile set on the committed tree (c51's was taken with the identity skip still in the straight-line kernels), and all workloads once more
bash profiles/run_profile.sh r04_hier --workload big-scene --traversal hier > gpurun_out/c63_prof.log 2>&1
bash profiles/workloads.sh > gpurun_out/c63_workloads.txt 2>&1
for w in "water-glass" "water-glass --traversal hier" "aquarium --traversal hier" "big-soup --samples 64" "big-mesh --samples 64" "mirror --traversal kd" "cows --traversal kd"; do
python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload $w 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('--workload %-40s %9.1f Mray/s %8.2f ms/frame' % ('$w', d['value'], d['ms_per_step']))" >> gpurun_out/c63_workloads.txt
done
cat gpurun_out/c63_workloads.txt
