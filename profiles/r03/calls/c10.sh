# round 3, call 10: the suite with the fork / join tests, k-d culls with folded margins, the texture routine inline, lane counters with and without forking
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/c10_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c10_pytest.log
run() { timeout 300 python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 "$@" 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-28s %-40s %9.1f Mray/s %9.2f ms/frame  %s' % ('$TAG', '$*', d['value'], d['ms_per_step'], d['roofline']['kernel'][5:60]))"; }
TAG=main; for wl in "big-scene" "big-scene --traversal hier" "big-scene --traversal kd" "mirror" "mirror --traversal kd" "cows" "cows --traversal kd" "aquarium" "aquarium --traversal hier"; do run --workload $wl; done > gpurun_out/c10_workloads.log 2>&1
cp portrayer_amd/libportrayer_hip.so /tmp/keep.so
cp build/variants/mapsinl/libportrayer_hip.so portrayer_amd/libportrayer_hip.so; TAG="mapsinl"; for wl in "aquarium" "aquarium --traversal hier"; do run --workload $wl; done >> gpurun_out/c10_workloads.log 2>&1
cp /tmp/keep.so portrayer_amd/libportrayer_hip.so
for f in 0 1; do export PORTRAYER_FORK=$f; echo "== PORTRAYER_FORK=$f"; timeout 600 bash profiles/pmc_quick.sh "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" --no-extras --workload aquarium; done > gpurun_out/c10_fork_pmc.log 2>&1
unset PORTRAYER_FORK
