run() { python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload $* 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-50s %9.1f Mray/s %9.2f ms/frame' % (sys.argv[1], d['value'], d['ms_per_step']))" "$*"; }
( for bm in 32 16 8 4 2; do echo "== PORTRAYER_BATCH_MAX=$bm"; export PORTRAYER_BATCH_MAX=$bm
  run big-scene; run "big-scene --traversal kd"; run "big-scene --traversal hier";  run "big-soup --samples 64"; run big-soup; run big-mesh; run cows; run "big-scene --share 8"
done
unset PORTRAYER_BATCH_MAX; export PORTRAYER_FINE_QUEUES=64; echo "== PORTRAYER_FINE_QUEUES=64"
  run big-scene; run "big-scene --traversal kd"; run "big-scene --traversal hier";  run "big-soup --samples 64"; run big-soup; run big-mesh; run cows; run "big-scene --share 8"
) > gpurun_out/c31_batch.log 2>&1
