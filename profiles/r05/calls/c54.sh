# round 5, call 54: pt_cand_end without branches (modes 1, 3): parity, A/B
OBJ="1=build/diag/m1_candend.o 3=build/diag/m3_candend.o"
bash profiles/r05/with_objs.sh "$OBJ" timeout 1200 python -m pytest tests/test_gpu_render_parity.py -m gpu -q -x --timeout=900 > gpurun_out/c54_pytest.log 2>&1; tail -1 gpurun_out/c54_pytest.log
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-72s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
for rep in 1 2; do
for a in "--workload big-scene" "--workload big-soup --samples 64" "--workload big-mesh --samples 64" "--workload cows" "--workload mirror"; do
  bash profiles/r05/with_objs.sh "1=build/diag/m1_before.o 3=build/diag/m3_before.o" python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "before $a"
  bash profiles/r05/with_objs.sh "$OBJ" python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "range end without branches $a"
done; done > gpurun_out/c54_candend.txt 2>&1
cat gpurun_out/c54_candend.txt
