run() { python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload $* 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-50s %9.1f Mray/s %9.2f ms/frame' % (sys.argv[1], d['value'], d['ms_per_step']))" "$*"; }
python -m pytest tests -m gpu -x -q > gpurun_out/c25_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c25_pytest.log
bash profiles/workloads.sh --no-extras > gpurun_out/c25_workloads.log 2>&1
( for park in 0 1; do echo "== PORTRAYER_PARK=$park"; export PORTRAYER_PARK=$park
  run aquarium; run "aquarium --traversal hier"; run "aquarium --samples 64 --steps 2"; run mirror; run "mirror --traversal hier"; run "mirror --traversal kd"
done ) > gpurun_out/c25_park.log 2>&1
