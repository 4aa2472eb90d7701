run() { python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload $* 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-50s %9.1f Mray/s %9.2f ms/frame' % (sys.argv[1], d['value'], d['ms_per_step']))" "$*"; }
( for fq in 0 1 8 64; do echo "== PORTRAYER_FINE_QUEUES=$fq"; export PORTRAYER_FINE_QUEUES=$fq
  run aquarium; run "aquarium --traversal hier"; run "aquarium --samples 64 --steps 2"; run mirror; run "mirror --traversal hier"; run big-scene; run cows
done ) > gpurun_out/c30_fine.log 2>&1
unset PORTRAYER_FINE_QUEUES
python -m pytest tests -m gpu -x -q > gpurun_out/c30_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c30_pytest.log
