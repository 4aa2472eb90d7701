run() { python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload $* 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-50s %9.1f Mray/s %9.2f ms/frame  nodes/ray %.2f tri/ray %.2f' % (sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['per_ray']['inner_nodes'], d['roofline']['per_ray']['triangle_tests']))" "$*"; }
( for cm in mesh plain; do echo "== PORTRAYER_COLLAPSE=$cm"; export PORTRAYER_COLLAPSE=$cm
  run "big-soup --samples 64"; run big-mesh; run cows; run mirror; run aquarium; run "mirror --traversal kd"; run "big-soup --samples 64 --traversal hier"
done ) > gpurun_out/c43_collapse.log 2>&1
unset PORTRAYER_COLLAPSE
python -m pytest tests -m gpu -x -q > gpurun_out/c43_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c43_pytest.log
