run() { python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload $* 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-50s %9.1f Mray/s %9.3f ms/frame' % (sys.argv[1], d['value'], d['ms_per_step']))" "$*"; }
( for bl in 1 2 3 4 6 8; do echo "== PORTRAYER_BLAS_LEAF=$bl"; export PORTRAYER_BLAS_LEAF=$bl
run "big-soup --samples 64"; run "big-mesh"; run "cows"; run "mirror"; run "aquarium"
done; unset PORTRAYER_BLAS_LEAF
for tl in 1 2 4; do echo "== PORTRAYER_TLAS_LEAF=$tl"; export PORTRAYER_TLAS_LEAF=$tl; run "big-scene"; run "mirror"; run "big-mesh"; done; unset PORTRAYER_TLAS_LEAF
for r in 8 32 64; do echo "== PORTRAYER_PLOC_RADIUS=$r"; export PORTRAYER_PLOC_RADIUS=$r; run "big-soup --samples 64"; done ) > gpurun_out/c83.log 2>&1
