# round 3, call 30: hierarchical semantics: paths through hier_rec in pt_hit_surface only (c29's fetch of the last level with the node record cost registers and 9 %: dropped)
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/c30_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c30_pytest.log
run() { timeout 300 python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 "$@" 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-44s %9.1f Mray/s %9.2f ms/frame  %s' % ('$*', d['value'], d['ms_per_step'], d['roofline']['kernel'][5:64]))"; }
for wl in "big-scene --traversal hier" "mirror --traversal hier" "cows --traversal hier" "aquarium --traversal hier" "water-glass --traversal hier" "big-soup --samples 64 --traversal hier" "big-scene" "mirror"; do run --workload $wl; done > gpurun_out/c30_hier.log 2>&1
timeout 600 python3 tests/fuzz_gpu_parity.py 16000 60 > gpurun_out/c30_fuzz.log 2>&1
