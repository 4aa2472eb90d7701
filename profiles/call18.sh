python -m pytest tests -m gpu -x -q 2>&1 | tail -2 > gpurun_out/c18_pytest.log
for wl in "big-scene --traversal kd" "mirror --traversal kd" "cows --traversal kd" "aquarium"; do
python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload $wl 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-40s %9.1f Mray/s %9.2f ms/frame' % ('$wl', d['value'], d['ms_per_step']))"
done > gpurun_out/c18_kd.log 2>&1
