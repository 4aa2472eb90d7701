# round 3, call 33: a line-against-box test in local space before the primitive's own test (-DPT_LEAF_CULL)
run() { timeout 300 python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 "$@" 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-10s %-44s %9.1f Mray/s %9.2f ms/frame  %s' % ('$TAG', '$*', d['value'], d['ms_per_step'], d['roofline']['kernel'][5:64]))"; }
cp portrayer_amd/libportrayer_hip.so /tmp/keep.so
cp build/variants/cull/libportrayer_hip.so portrayer_amd/libportrayer_hip.so
TAG=cull; for wl in "big-scene" "big-scene --traversal hier" "mirror" "cows" "aquarium" "primitives"; do run --workload $wl; done > gpurun_out/c33_cull.log 2>&1
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/c33_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c33_pytest.log
timeout 600 python3 tests/fuzz_gpu_parity.py 22000 60 > gpurun_out/c33_fuzz.log 2>&1
cp /tmp/keep.so portrayer_amd/libportrayer_hip.so
