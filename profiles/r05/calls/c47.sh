# round 5, call 47: wave-count policy re-checked on the new step (mesh scenes 4 / 5, headline 5 / 6), one rank's share, two frames in flight
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-72s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
for cfg in "PORTRAYER_WAVES=4" "PORTRAYER_WAVES=5" "X=0"; do
for a in "--workload big-soup --samples 64" "--workload big-mesh --samples 64" "--workload cows" "--workload mirror" "--workload big-scene"; do
  env $cfg python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "$cfg $a"
done; done > gpurun_out/c47_waves.txt 2>&1
for a in "--workload big-scene --share 8 --share-rank 0" "--workload big-scene --share 8 --share-rank 0 --overlap" "--workload big-scene --share 8 --share-rank 5" "--workload big-scene --overlap"; do
  python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 4 $a 2>/dev/null | line "$a"
done >> gpurun_out/c47_waves.txt 2>&1
cat gpurun_out/c47_waves.txt
