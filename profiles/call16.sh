python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/c16_pytest.log
for c in 1 2 4 8; do echo "== PORTRAYER_LANE_CHUNKS=$c"; for wl in "big-scene" "big-soup --samples 64" "mirror" "aquarium --samples 64 --steps 1" "big-scene --traversal kd" "big-scene --traversal hier"; do
PORTRAYER_LANE_CHUNKS=$c python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload $wl 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-40s %9.1f Mray/s %9.2f ms/frame' % ('$wl', d['value'], d['ms_per_step']))"
done; done > gpurun_out/c16_chunks.log 2>&1
