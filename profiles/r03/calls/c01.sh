# round 3, call 1: the straight-line kernel (pt_render_simple.h) against the interpreter kernel it replaces (build with -DPT_KEEP_INTERP)
python -m pytest tests -m gpu -x -q > gpurun_out/c01_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c01_pytest.log
echo "== straight-line kernel" > gpurun_out/c01_workloads.log
bash profiles/workloads.sh --no-extras >> gpurun_out/c01_workloads.log 2>&1
echo "== interpreter kernel (PORTRAYER_INTERP=1)" >> gpurun_out/c01_workloads.log
PORTRAYER_INTERP=1 bash profiles/workloads.sh --no-extras >> gpurun_out/c01_workloads.log 2>&1
for wl in "big-soup --samples 64" "big-mesh --samples 64" "big-scene --share 8" "primitives"; do
for e in 0 1; do
PORTRAYER_INTERP=$e python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload $wl 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('interp=$e %-60s %9.1f Mray/s %9.2f ms/frame' % ('$wl', d['value'], d['ms_per_step']))"
done; done >> gpurun_out/c01_workloads.log 2>&1
