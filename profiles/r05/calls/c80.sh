# round 5, call 80: the candidate's range end from two lane masks kept for the instance walk (mode 1): parity, A/B
bash profiles/r05/with_objs.sh "1=build/diag/m1_masks.o" timeout 1200 python -m pytest tests/test_gpu_render_parity.py tests/test_gpu_timed_sizes.py -m gpu -q -x --timeout=900 > gpurun_out/c80_pytest.log 2>&1; tail -1 gpurun_out/c80_pytest.log
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-72s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
for rep in 1 2; do for a in "--workload big-soup --samples 64" "--workload big-mesh --samples 64" "--workload cows" "--workload mirror"; do
  python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "shipped $a"
  bash profiles/r05/with_objs.sh "1=build/diag/m1_masks.o" python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "range end from masks $a"
done; done > gpurun_out/c80_cand_masks.txt 2>&1
cat gpurun_out/c80_cand_masks.txt
