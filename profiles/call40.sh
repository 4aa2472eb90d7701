for wl in "big-scene --width 3840 --height 2160 --samples 256" "big-scene --width 3840 --height 2160 --samples 256 --share 8" "big-scene --share 2" "big-scene --share 4" "big-scene --share 8" "cows --traversal hier" "cows --traversal kd" "mirror --traversal kd" "big-soup --samples 64" "big-soup --samples 64 --traversal hier" "aquarium --traversal hier" "aquarium --samples 64 --steps 2" "big-mesh --samples 64"; do
python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload $wl 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-70s %9.1f Mray/s %9.2f ms/frame' % ('$wl', d['value'], d['ms_per_step']))"
done > gpurun_out/c40_more.log 2>&1
PORTRAYER_BUILD=host python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload big-soup --samples 64 2>&1 | tail -1 | cut -c1-200 >> gpurun_out/c40_more.log
python bench.py --steps 10 --warmup 3 > gpurun_out/c40_bench.log 2>&1; echo "rc $?" >> gpurun_out/c40_bench.log
