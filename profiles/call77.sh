python -m pytest tests -m gpu -x -q > gpurun_out/c77_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c77_pytest.log
bash profiles/workloads.sh --no-extras > gpurun_out/c77_workloads.log 2>&1
for wl in "big-scene --width 3840 --height 2160 --samples 256" "big-scene --width 3840 --height 2160 --samples 256 --share 8" "big-scene --share 2" "big-scene --share 4" "big-scene --share 8" "cows --traversal hier" "cows --traversal kd" "mirror --traversal kd" "big-soup --samples 64" "big-soup --samples 64 --traversal hier" "aquarium --traversal hier" "aquarium --samples 64 --steps 2" "big-mesh --samples 64"; do
python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload $wl 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-70s %9.1f Mray/s %9.2f ms/frame' % ('$wl', d['value'], d['ms_per_step']))"
done > gpurun_out/c77_more.log 2>&1
PORTRAYER_BUILD=host python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload big-soup --samples 64 2>&1 | tail -1 | cut -c1-200 >> gpurun_out/c77_more.log
