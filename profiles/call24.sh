run() { python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload $* 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-50s %9.1f Mray/s %9.2f ms/frame' % (sys.argv[1], d['value'], d['ms_per_step']))" "$*"; }
export PORTRAYER_PARK=1
bash profiles/timeline.sh "--workload aquarium" "--workload aquarium --samples 64" "--workload mirror" "--workload big-scene" "--workload big-soup --samples 64" > gpurun_out/c24_timeline.log 2>&1
( for st in 1 golden 7919 64 ; do echo "== PORTRAYER_ITEM_STRIDE=$st"; export PORTRAYER_ITEM_STRIDE=$st
  run aquarium; run "aquarium --samples 64 --steps 2"; run mirror; run big-scene; run "big-soup --samples 64"
done ) > gpurun_out/c24_stride.log 2>&1
export PORTRAYER_ITEM_STRIDE=golden
bash profiles/timeline.sh "--workload aquarium" >> gpurun_out/c24_timeline.log 2>&1
