run() { python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload $* 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-50s %9.1f Mray/s %9.2f ms/frame' % (sys.argv[1], d['value'], d['ms_per_step']))" "$*"; }
for park in 0 1 2; do for kb in 52 80; do
  echo "== PORTRAYER_PARK=$park PORTRAYER_LDS_BUDGET_KB=$kb"
  export PORTRAYER_PARK=$park PORTRAYER_LDS_BUDGET_KB=$kb
  run aquarium; run mirror; run "mirror --traversal hier"; run "aquarium --traversal hier"
done; done > gpurun_out/c21_park.log 2>&1
unset PORTRAYER_LDS_BUDGET_KB
for park in 1 2; do
PORTRAYER_PARK=$park python -m pytest tests -m gpu -x -q -k "not config_size" > gpurun_out/c21_pytest_park$park.log 2>&1; echo "pytest rc $?" >> gpurun_out/c21_pytest_park$park.log
done
PORTRAYER_PARK=1 PORTRAYER_LDS_BUDGET_KB=80 timeout 600 python tests/fuzz_gpu_parity.py 7000 40 > gpurun_out/c21_fuzz.log 2>&1
