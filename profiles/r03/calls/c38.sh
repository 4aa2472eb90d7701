# round 3, call 38: why the hierarchical semantics did not finish at 6 waves per SIMD in c37 (each attempt under its own short timeout)
cp portrayer_amd/libportrayer_hip.so /tmp/keep.so
cp build/variants/w6/libportrayer_hip.so portrayer_amd/libportrayer_hip.so
try() { echo "== $*"; ( timeout 40 python3 bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 0 --workload big-scene --traversal hier "$@" 2>&1 | tail -2 | cut -c1-300 ); echo "rc $?"; }
{
PORTRAYER_LDS_BUDGET_KB=26 try --width 320 --height 180 --samples 64
PORTRAYER_LDS_BUDGET_KB=31 try --width 320 --height 180 --samples 64
PORTRAYER_LDS_BUDGET_KB=26 PORTRAYER_FINE_QUEUES=0 try --width 320 --height 180 --samples 64
PORTRAYER_LDS_BUDGET_KB=26 try --width 1920 --height 1080 --samples 64
} > gpurun_out/c38_hier6.log 2>&1
cp /tmp/keep.so portrayer_amd/libportrayer_hip.so
