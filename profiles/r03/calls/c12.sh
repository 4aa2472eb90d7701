# round 3, call 12: 3 against 4 waves per SIMD with the straight-line kernel, workload by workload
run() { timeout 300 python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 "$@" 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-12s %-44s %9.1f Mray/s %9.2f ms/frame  %s' % ('$TAG', '$*', d['value'], d['ms_per_step'], d['roofline']['kernel'][5:60]))"; }
for wv in 3 4; do export PORTRAYER_WAVES=$wv; TAG="waves=$wv"; for wl in "big-scene" "big-scene --traversal hier" "cows" "cows --traversal hier" "primitives" "big-soup --samples 64" "big-soup --samples 64 --traversal hier" "big-mesh --samples 64" "big-scene --width 3840 --height 2160 --samples 256"; do run --workload $wl; done; done > gpurun_out/c12_waves.log 2>&1
unset PORTRAYER_WAVES
for wv in 3 4; do export PORTRAYER_KD_WAVES=$wv; TAG="kdwaves=$wv"; for wl in "big-scene --traversal kd" "primitives --traversal kd"; do run --workload $wl; done; done >> gpurun_out/c12_waves.log 2>&1
