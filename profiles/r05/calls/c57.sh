# round 5, call 57: the headline's leaf tests with the record arrays' addresses pinned (they are re-read from the argument block per test): parity, A/B, alternating
P="3=build/diag/m3_pinleaf.o 6=build/diag/m6_pinleaf.o"
S="3=build/diag/m3_same.o 6=build/diag/m6_same.o"
bash profiles/r05/with_objs.sh "$P" timeout 900 python -m pytest tests/test_gpu_render_parity.py -m gpu -q -x -k "example_matches or random or headline" --timeout=900 > gpurun_out/c57_pytest.log 2>&1; tail -1 gpurun_out/c57_pytest.log
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-72s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
for rep in 1 2 3; do
for a in "--workload big-scene" "--workload big-scene --traversal hier"; do
  bash profiles/r05/with_objs.sh "$S" python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 4 $a 2>/dev/null | line "as shipped $a"
  bash profiles/r05/with_objs.sh "$P" python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 4 $a 2>/dev/null | line "leaf arrays pinned $a"
done; done > gpurun_out/c57_pin_leaf.txt 2>&1
cat gpurun_out/c57_pin_leaf.txt
