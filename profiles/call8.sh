python -m pytest tests -m gpu -x -q > gpurun_out/c8_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c8_pytest.log
for wv in 3 4; do for wl in "big-scene --traversal kd" "mirror --traversal kd" "cows --traversal kd"; do
PORTRAYER_WAVES=$wv python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload $wl 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-32s waves %s %9.1f Mray/s %9.2f ms/frame' % ('$wl', '$wv', d['value'], d['ms_per_step']))"
done; done > gpurun_out/c8_kd.log 2>&1
python -c "
import sys; sys.path.insert(0,'.')
from portrayer_amd import _hip as H
c = H.Context(0); print('copy bandwidth GB/s', c.copy_bandwidth(1<<30, 5))" >> gpurun_out/c8_kd.log 2>&1
