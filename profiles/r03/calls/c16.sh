# round 3, call 16: the straight-line kernel with a depth loop (scenes whose reflective materials are all opaque) against the interpreter
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/c16_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c16_pytest.log
run() { timeout 300 python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 "$@" 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-12s %-44s %9.1f Mray/s %9.2f ms/frame  %s' % ('$TAG', '$*', d['value'], d['ms_per_step'], d['roofline']['kernel'][5:64]))"; }
for ch in 1 0; do export PORTRAYER_CHAIN=$ch; TAG="chain=$ch"; for wl in "mirror" "mirror --traversal hier" "mirror --traversal kd" "mirror --samples 256"; do run --workload $wl; done; done > gpurun_out/c16_chain.log 2>&1
unset PORTRAYER_CHAIN
timeout 900 python3 tests/fuzz_gpu_parity.py 12000 150 > gpurun_out/c16_fuzz.log 2>&1
bash profiles/cycles.sh "--workload big-scene" "--workload big-scene --traversal hier" "--workload mirror" "--workload cows" "--workload aquarium" > gpurun_out/c16_cycles.log 2>&1
