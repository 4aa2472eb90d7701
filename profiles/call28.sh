run() { python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload $* 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-50s %9.1f Mray/s %9.2f ms/frame' % (sys.argv[1], d['value'], d['ms_per_step']))" "$*"; }
( for st in 1 golden; do for bm in 1 4 32; do echo "== PORTRAYER_ITEM_STRIDE=$st PORTRAYER_BATCH_MAX=$bm"; export PORTRAYER_ITEM_STRIDE=$st PORTRAYER_BATCH_MAX=$bm
  run aquarium; run mirror; run big-scene; run "big-soup --samples 64"; run cows
done; done ) > gpurun_out/c28_batch.log 2>&1
