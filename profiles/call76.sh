run() { python3 bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 --workload $* 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-60s %9.1f Mray/s %9.3f ms/frame' % (sys.argv[1], d['value'], d['ms_per_step']))" "$*"; }
( for cfg in "" "PORTRAYER_FINE_QUEUES=64"; do echo "== $cfg"; for kv in $cfg; do export $kv; done
run "big-scene"; run "big-scene --traversal hier"; run "big-scene --traversal kd"; run "big-soup --samples 64"; run "big-soup"; run "big-mesh"; run "cows --traversal hier"; run "cows --traversal kd"; run "big-scene --width 3840 --height 2160 --samples 256 --steps 2"; run "big-scene --width 3840 --height 2160 --samples 256 --share 8"; run "primitives"
unset PORTRAYER_FINE_QUEUES; done ) > gpurun_out/c76_fine.log 2>&1
