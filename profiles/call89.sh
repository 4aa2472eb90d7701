run() { python3 bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 --workload $* 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-40s %9.1f Mray/s %9.3f ms/frame' % (sys.argv[1], d['value'], d['ms_per_step']))" "$*"; }
( for q in 4 8 16 32 64; do echo "== PORTRAYER_FINE_QUEUES=$q"; export PORTRAYER_FINE_QUEUES=$q
run "big-scene"; run "big-scene --share 8"; run "cows"; run "mirror"; run "big-soup --samples 64"; done ) > gpurun_out/c89.log 2>&1
