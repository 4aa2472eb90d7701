# round 5, call 68: the k-d walk at 4 against 5 waves per SIMD after the common-case split
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-72s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
for rep in 1 2; do for cfg in "PORTRAYER_KD_WAVES=4" "PORTRAYER_KD_WAVES=5" "PORTRAYER_KD_WAVES=3"; do
  env $cfg python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 --workload big-scene --traversal kd 2>/dev/null | line "$cfg big-scene kd"
done; done > gpurun_out/c68_kd_waves.txt 2>&1
cat gpurun_out/c68_kd_waves.txt
