run() { python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload $* 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-50s %9.1f Mray/s %9.2f ms/frame' % (sys.argv[1], d['value'], d['ms_per_step']))" "$*"; }
python -m pytest tests -m gpu -x -q -k "hier or HIER or example or config" > gpurun_out/c37_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c37_pytest.log
( for v in 0 1; do if [ $v = 1 ]; then export PORTRAYER_HIER_NO_SKIP=1; fi; echo "== NO_SKIP=$v"
  run "big-scene --traversal hier"; run "mirror --traversal hier"; run "cows --traversal hier"; run "aquarium --traversal hier"; run "big-soup --samples 64 --traversal hier"
done ) > gpurun_out/c37_hier.log 2>&1
unset PORTRAYER_HIER_NO_SKIP
timeout 1200 python tests/fuzz_gpu_parity.py 13000 120 > gpurun_out/c37_fuzz.log 2>&1
