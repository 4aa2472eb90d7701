run() { python3 bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 2 --workload $* 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-40s %9.1f Mray/s %9.3f ms/frame  nodes/ray %.2f prim/ray %.3f' % (sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['per_ray']['inner_nodes'], d['roofline']['per_ray']['primitive_tests']))" "$*"; }
( for m in binned sweep; do echo "== PORTRAYER_SAH=$m"; export PORTRAYER_SAH=$m
run "big-scene"; run "big-scene --traversal hier"; run "mirror"; run "cows"; run "aquarium"; done ) > gpurun_out/c84.log 2>&1
