run() { python3 bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 3 --workload $* 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-50s %9.1f Mray/s %9.3f ms/frame' % (sys.argv[1], d['value'], d['ms_per_step']))" "$*"; }
( for cfg in "" "PORTRAYER_FINE_QUEUES=64" "PORTRAYER_BATCH_MAX=4" "PORTRAYER_BATCH_MAX=2" "PORTRAYER_WAVES=3" "PORTRAYER_WAVES=3 PORTRAYER_FINE_QUEUES=64"; do echo "== $cfg"; for kv in $cfg; do export $kv; done
run "big-scene --share 8"; run "big-scene --share 4"; run "big-scene --share 2"; run "cows"
unset PORTRAYER_FINE_QUEUES PORTRAYER_BATCH_MAX PORTRAYER_WAVES; done ) > gpurun_out/c75_share.log 2>&1
