run() { python3 bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 2 --workload $* 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-50s %9.1f Mray/s %9.3f ms/frame' % (sys.argv[1], d['value'], d['ms_per_step']))" "$*"; }
( for cfg in "" "PORTRAYER_PARK=0" "PORTRAYER_PARK=0 PORTRAYER_WAVES=4"; do echo "== $cfg"; for kv in $cfg; do export $kv; done
run "mirror"; run "mirror --traversal hier"; run "aquarium"; run "aquarium --traversal hier"
unset PORTRAYER_PARK PORTRAYER_WAVES; done ) > gpurun_out/c82.log 2>&1
