run() { python3 bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 --workload $* 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-40s %9.1f Mray/s %9.2f ms/frame' % (sys.argv[1], d['value'], d['ms_per_step']))" "$*"; }
( for i in 1 2 3; do
echo "== default"; unset PORTRAYER_WAVES; run big-scene; run "big-scene --traversal hier"; run "big-soup --samples 64"
echo "== WAVES=3"; export PORTRAYER_WAVES=3; run big-scene; run "big-scene --traversal hier"
echo "== WAVES=4"; export PORTRAYER_WAVES=4; run big-scene; run "big-scene --traversal hier"
done ) > gpurun_out/c67.log 2>&1
