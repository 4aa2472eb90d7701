# round 3, call 9: fork / join (N1) after the ticket fix, under timeouts; branch-free primitive tests; reflective-kernel variants
timeout 240 python -m pytest tests/test_gpu_textures.py tests/test_examples_extra.py -m gpu -x -q -k "transmission or water or glass or recursion" > gpurun_out/c09_fork_smoke.log 2>&1
rc=$?; echo "smoke rc $rc" >> gpurun_out/c09_fork_smoke.log
if [ $rc -ne 0 ]; then export PORTRAYER_FORK=0; echo "fork disabled for the rest of the call" >> gpurun_out/c09_fork_smoke.log; fi
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/c09_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c09_pytest.log
run() { timeout 300 python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 "$@" 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-28s %-40s %9.1f Mray/s %9.2f ms/frame  %s' % ('$TAG', '$*', d['value'], d['ms_per_step'], d['roofline']['kernel'][5:60]))"; }
TAG=main; for wl in "big-scene" "big-scene --traversal hier" "big-scene --traversal kd" "mirror" "cows" "big-soup --samples 64" "big-mesh --samples 64"; do run --workload $wl; done > gpurun_out/c09_workloads.log 2>&1
if [ $rc -eq 0 ]; then
for f in 1 0; do export PORTRAYER_FORK=$f; TAG="fork=$f"; run --workload aquarium; run --workload aquarium --samples 64 --steps 2; run --workload aquarium --traversal hier; done > gpurun_out/c09_fork.log 2>&1
unset PORTRAYER_FORK
fi
cp portrayer_amd/libportrayer_hip.so /tmp/keep.so
for v in interp2 powinl mapsinl; do cp build/variants/$v/libportrayer_hip.so portrayer_amd/libportrayer_hip.so; TAG="$v"; run --workload aquarium; run --workload mirror; run --workload big-scene; run --workload cows; done > gpurun_out/c09_variants.log 2>&1
cp /tmp/keep.so portrayer_amd/libportrayer_hip.so
timeout 600 python3 tests/fuzz_gpu_parity.py 9000 120 > gpurun_out/c09_fuzz.log 2>&1
