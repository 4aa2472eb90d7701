# round 5, call 60: leaf sizes of the mesh trees re-checked on the cheaper step
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-72s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
for cfg in "PORTRAYER_BLAS_LEAF=1" "PORTRAYER_BLAS_LEAF=2" "PORTRAYER_BLAS_LEAF=3" "PORTRAYER_BLAS_LEAF=4"; do
for a in "--workload big-soup --samples 64" "--workload big-mesh --samples 64" "--workload cows" "--workload mirror"; do
  env $cfg python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "$cfg $a"
done; done > gpurun_out/c60_leaf_sizes.txt 2>&1
cat gpurun_out/c60_leaf_sizes.txt
