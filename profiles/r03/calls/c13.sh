# round 3, call 13: 5 waves per SIMD (96 registers, 31 KB of LDS per block) for the mesh-free kernels; the suite with the new wave policy
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/c13_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c13_pytest.log
run() { timeout 300 python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 "$@" 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-12s %-44s %9.1f Mray/s %9.2f ms/frame  %s' % ('$TAG', '$*', d['value'], d['ms_per_step'], d['roofline']['kernel'][5:60]))"; }
TAG=main; for wl in "big-scene" "big-scene --traversal hier" "big-scene --traversal kd"; do run --workload $wl; done > gpurun_out/c13_w5.log 2>&1
cp portrayer_amd/libportrayer_hip.so /tmp/keep.so
cp build/variants/w5/libportrayer_hip.so portrayer_amd/libportrayer_hip.so
export PORTRAYER_LDS_BUDGET_KB=31
TAG="5 waves"; for wl in "big-scene" "big-scene --traversal hier" "big-scene --traversal kd" "big-scene --width 3840 --height 2160 --samples 256"; do run --workload $wl; done >> gpurun_out/c13_w5.log 2>&1
PORTRAYER_WAVES=4 run --workload big-scene --traversal hier >> gpurun_out/c13_w5.log 2>&1
unset PORTRAYER_LDS_BUDGET_KB
cp /tmp/keep.so portrayer_amd/libportrayer_hip.so
