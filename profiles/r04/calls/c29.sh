# round 4, call 29: 5 waves per SIMD for the kernels of scenes with mesh instances (96 registers, 57 spilled): the 1.25 M-triangle scenes wait for node fetches (3 -> 4 waves: +18 %)
run() { name=$1; shift
  env "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']['per_ray']
print('%-44s %9.1f Mray/s %8.3f ms  %s' % ('$name', d['value'], d['ms_per_step'], d['roofline']['kernel'][5:60]))"
}
B="python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1"
for rep in 1 2; do
run "big-soup x64, 4 waves" X=1 $B --workload big-soup --samples 64 >> gpurun_out/c29.txt
run "big-soup x64, 5 waves" PORTRAYER_WAVES=5 $B --workload big-soup --samples 64 >> gpurun_out/c29.txt
done
run "big-soup x64, 5 waves, host SAH" PORTRAYER_WAVES=5 PORTRAYER_BUILD=host $B --workload big-soup --samples 64 >> gpurun_out/c29.txt
run "big-mesh x64, 4 waves" X=1 $B --workload big-mesh --samples 64 >> gpurun_out/c29.txt
run "big-mesh x64, 5 waves" PORTRAYER_WAVES=5 $B --workload big-mesh --samples 64 >> gpurun_out/c29.txt
run "big-soup x64 hier, 4 waves" X=1 $B --workload big-soup --samples 64 --traversal hier >> gpurun_out/c29.txt
run "big-soup x64 hier, 5 waves" PORTRAYER_WAVES=5 $B --workload big-soup --samples 64 --traversal hier >> gpurun_out/c29.txt
run "cows, 4 waves" X=1 $B --workload cows >> gpurun_out/c29.txt
run "cows, 5 waves" PORTRAYER_WAVES=5 $B --workload cows >> gpurun_out/c29.txt
