# round 5, call 19: the restructured k-d walk (children in the octant's order): nodes per ray and instruction counts
python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload big-scene --traversal kd 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('kd big-scene', d['value'], d['ms_per_step'], d['roofline']['per_ray'])"
cd /tmp; export TMPDIR=/tmp
timeout 300 bash $GRAFT_REPO_ROOT/profiles/pmc_quick.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" --no-extras --workload big-scene --traversal kd
