# round 3, call 8: fork / join of refracted subtrees in the reflective kernel (N1), and what the reflective kernel's scratch traffic costs
python -m pytest tests -m gpu -q -x > gpurun_out/c08_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c08_pytest.log
run() { python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 "$@" 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-28s %-40s %9.1f Mray/s %9.2f ms/frame  %s' % ('$TAG', '$*', d['value'], d['ms_per_step'], d['roofline']['kernel'][5:60]))"; }
for f in 1 0; do export PORTRAYER_FORK=$f; TAG="fork=$f"; run --workload aquarium; run --workload aquarium --samples 64 --steps 2; run --workload aquarium --traversal hier; run --workload mirror; done > gpurun_out/c08_fork.log 2>&1
unset PORTRAYER_FORK
cp portrayer_amd/libportrayer_hip.so /tmp/keep.so
for v in interp2 powinl mapsinl; do cp build/variants/$v/libportrayer_hip.so portrayer_amd/libportrayer_hip.so; TAG="$v"; run --workload aquarium; run --workload mirror; run --workload big-scene; run --workload cows; done > gpurun_out/c08_variants.log 2>&1
cp build/variants/interp2/libportrayer_hip.so portrayer_amd/libportrayer_hip.so; TAG="interp2 lds80"; PORTRAYER_LDS_BUDGET_KB=80 run --workload aquarium >> gpurun_out/c08_variants.log 2>&1
cp /tmp/keep.so portrayer_amd/libportrayer_hip.so
