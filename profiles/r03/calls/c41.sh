# round 3, call 41: an instantiation of the hierarchical semantics for scenes with Mesh instances but no KDMesh trees (mode 8)
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/c41_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c41_pytest.log
run() { timeout 300 python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 "$@" 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-14s %-50s %9.1f Mray/s %9.2f ms/frame  %s' % ('$TAG', '$*', d['value'], d['ms_per_step'], d['roofline']['kernel'][5:64]))"; }
TAG=default; for wl in "mirror --traversal hier" "cows --traversal hier" "big-soup --samples 64 --traversal hier" "big-mesh --samples 64 --traversal hier" "big-mesh --traversal hier" "aquarium --traversal hier" "big-scene --traversal hier"; do run --workload $wl; done > gpurun_out/c41_hier.log 2>&1
export PORTRAYER_CHAIN_WAVES=4; TAG="chain waves 4"; run --workload mirror --traversal hier >> gpurun_out/c41_hier.log 2>&1; unset PORTRAYER_CHAIN_WAVES
export PORTRAYER_WAVES=4; TAG="waves 4"; for wl in "cows --traversal hier" "cows"; do run --workload $wl; done >> gpurun_out/c41_hier.log 2>&1; unset PORTRAYER_WAVES
timeout 600 python3 tests/fuzz_gpu_parity.py 33000 60 > gpurun_out/c41_fuzz.log 2>&1
