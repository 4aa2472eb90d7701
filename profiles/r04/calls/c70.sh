# round 4, call 70: the wave-count policy swept once more on the final kernels (scenes with meshes; k-d)
run() { name=$1; shift
  env "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-44s %9.1f Mray/s %8.3f ms  %s' % ('$name', d['value'], d['ms_per_step'], d['roofline']['kernel'][5:62]))"
}
B="python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1"
for wv in 3 4 5; do
run "cows WAVES=$wv" PORTRAYER_WAVES=$wv $B --workload cows
run "cows hier WAVES=$wv" PORTRAYER_WAVES=$wv $B --workload cows --traversal hier
run "big-mesh x16 WAVES=$wv" PORTRAYER_WAVES=$wv $B --workload big-mesh
run "big-soup x16 WAVES=$wv" PORTRAYER_WAVES=$wv $B --workload big-soup
done > gpurun_out/c70_waves.txt 2>&1
for wv in 3 4; do
run "mirror CHAIN_WAVES=$wv" PORTRAYER_CHAIN_WAVES=$wv $B --workload mirror
run "mirror kd KD_WAVES=$wv" PORTRAYER_KD_WAVES=$wv $B --workload mirror --traversal kd
run "cows kd KD_WAVES=$wv" PORTRAYER_KD_WAVES=$wv $B --workload cows --traversal kd
run "big-scene kd KD_WAVES=$wv" PORTRAYER_KD_WAVES=$wv $B --workload big-scene --traversal kd
done >> gpurun_out/c70_waves.txt 2>&1
run "big-scene kd KD_WAVES=5" PORTRAYER_KD_WAVES=5 $B --workload big-scene --traversal kd >> gpurun_out/c70_waves.txt 2>&1
cat gpurun_out/c70_waves.txt
