# round 3, call 37: 6 waves per SIMD for the mesh-free straight-line kernels again, after the register work (80 registers, 27 spilled; 26 KB of LDS a block)
run() { timeout 300 python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 "$@" 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-10s %-54s %9.1f Mray/s %9.2f ms/frame  %s' % ('$TAG', '$*', d['value'], d['ms_per_step'], d['roofline']['kernel'][5:64]))"; }
TAG="5 waves"; for wl in "big-scene" "big-scene --traversal hier" "big-scene --width 3840 --height 2160 --samples 256" "big-scene --share 8"; do run --workload $wl; done > gpurun_out/c37_w6.log 2>&1
cp portrayer_amd/libportrayer_hip.so /tmp/keep.so
cp build/variants/w6/libportrayer_hip.so portrayer_amd/libportrayer_hip.so
export PORTRAYER_LDS_BUDGET_KB=26
TAG="6 waves"; for wl in "big-scene" "big-scene --traversal hier" "big-scene --width 3840 --height 2160 --samples 256" "big-scene --share 8"; do run --workload $wl; done >> gpurun_out/c37_w6.log 2>&1
unset PORTRAYER_LDS_BUDGET_KB
cp /tmp/keep.so portrayer_amd/libportrayer_hip.so
