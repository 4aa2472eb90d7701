# round 3, call 27: the interpreter kernels at 2 waves per SIMD (256 registers, nearly nothing spilled), forking off
run() { timeout 300 python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 "$@" 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-12s %-44s %9.1f Mray/s %9.2f ms/frame  %s' % ('$TAG', '$*', d['value'], d['ms_per_step'], d['roofline']['kernel'][5:64]))"; }
cp portrayer_amd/libportrayer_hip.so /tmp/keep.so
cp build/variants/w2/libportrayer_hip.so portrayer_amd/libportrayer_hip.so
for kb in 52 80; do export PORTRAYER_LDS_BUDGET_KB=$kb; TAG="2 waves $kb KB"; for wl in "aquarium" "aquarium --traversal hier" "aquarium --samples 64" "water-glass" "water-glass --traversal hier"; do run --workload $wl; done; done > gpurun_out/c27_w2.log 2>&1
unset PORTRAYER_LDS_BUDGET_KB
cp /tmp/keep.so portrayer_amd/libportrayer_hip.so
