# round 4, call 36: mesh-tree leaf size again, now that the tree step inside meshes is the octant-sorted one (17 instead of 25 instructions)
run() { name=$1; shift
  env "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-44s %9.1f Mray/s %8.3f ms' % ('$name', d['value'], d['ms_per_step']))"
}
B="python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1"
for l in 1 2 3 4; do
run "big-soup x64 BLAS_LEAF=$l" PORTRAYER_BLAS_LEAF=$l $B --workload big-soup --samples 64
run "big-mesh x64 BLAS_LEAF=$l" PORTRAYER_BLAS_LEAF=$l $B --workload big-mesh --samples 64
run "cows BLAS_LEAF=$l" PORTRAYER_BLAS_LEAF=$l $B --workload cows
run "mirror BLAS_LEAF=$l" PORTRAYER_BLAS_LEAF=$l $B --workload mirror
done > gpurun_out/c36_leaf.txt 2>&1
cat gpurun_out/c36_leaf.txt
