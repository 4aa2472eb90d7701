# round 3, call 11: k-d semantics at 4 waves per SIMD (mesh-free), texture routine inline in the FLAT_KDMESH kernels, 2-wave interpreter without forking
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/c11_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c11_pytest.log
run() { timeout 300 python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 "$@" 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-28s %-40s %9.1f Mray/s %9.2f ms/frame  %s' % ('$TAG', '$*', d['value'], d['ms_per_step'], d['roofline']['kernel'][5:60]))"; }
TAG=main; for wl in "big-scene --traversal kd" "aquarium" "aquarium --traversal hier" "aquarium --samples 64 --steps 2"; do run --workload $wl; done > gpurun_out/c11_workloads.log 2>&1
TAG="kd 4 waves"; PORTRAYER_KD_WAVES=4 run --workload big-scene --traversal kd >> gpurun_out/c11_workloads.log 2>&1
cp portrayer_amd/libportrayer_hip.so /tmp/keep.so
cp build/variants/interp2/libportrayer_hip.so portrayer_amd/libportrayer_hip.so; TAG="interp2 fork=0"; for wl in "aquarium" "aquarium --traversal hier"; do PORTRAYER_FORK=0 run --workload $wl; done >> gpurun_out/c11_workloads.log 2>&1
cp /tmp/keep.so portrayer_amd/libportrayer_hip.so
