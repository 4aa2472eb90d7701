# round 3, call 42: wave policy after c41 (plain-mesh scenes at 4 waves, chain kernel at 4 waves in the hierarchical mesh instantiation): suite, fuzz, workloads
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/c42_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c42_pytest.log
bash profiles/workloads.sh --no-extras > gpurun_out/c42_workloads.log 2>&1
run() { timeout 300 python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 "$@" 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-60s %9.1f Mray/s %9.2f ms/frame  %s' % ('$*', d['value'], d['ms_per_step'], d['roofline']['kernel'][5:64]))"; }
for wl in "cows --traversal hier" "big-soup --samples 64 --traversal hier" "big-mesh --samples 64 --traversal hier" "aquarium --traversal hier" "water-glass" "water-glass --traversal hier" "primitives" "primitives --traversal hier"; do run --workload $wl; done >> gpurun_out/c42_workloads.log 2>&1
timeout 600 python3 tests/fuzz_gpu_parity.py 34000 80 > gpurun_out/c42_fuzz.log 2>&1
