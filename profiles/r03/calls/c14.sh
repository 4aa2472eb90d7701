# round 3, call 14: the suite with the 5-wave kernels as default; 6 waves per SIMD as an experiment
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/c14_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c14_pytest.log
run() { timeout 300 python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 "$@" 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-12s %-44s %9.1f Mray/s %9.2f ms/frame  %s' % ('$TAG', '$*', d['value'], d['ms_per_step'], d['roofline']['kernel'][5:60]))"; }
TAG=main; for wl in "big-scene" "big-scene --traversal hier" "big-scene --traversal kd" "big-scene --share 8" "primitives"; do run --workload $wl; done > gpurun_out/c14_w6.log 2>&1
cp portrayer_amd/libportrayer_hip.so /tmp/keep.so
cp build/variants/w6/libportrayer_hip.so portrayer_amd/libportrayer_hip.so
export PORTRAYER_LDS_BUDGET_KB=26
TAG="6 waves"; for wl in "big-scene" "big-scene --traversal hier" "big-scene --width 3840 --height 2160 --samples 256"; do run --workload $wl; done >> gpurun_out/c14_w6.log 2>&1
unset PORTRAYER_LDS_BUDGET_KB
cp /tmp/keep.so portrayer_amd/libportrayer_hip.so
timeout 600 python3 tests/fuzz_gpu_parity.py 11000 100 > gpurun_out/c14_fuzz.log 2>&1
