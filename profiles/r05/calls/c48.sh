# round 5, call 48: the headline at 5 against 6 waves per SIMD on the new step (flat, hier, 4K x256), alternating; mesh scenes at 4 against 5 in the hierarchical semantics
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-72s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
W5="3=build/diag/w5.d/pt_render_m3.o 6=build/diag/w5.d/pt_render_m6.o api=build/diag/w5.d/pt_api.o"
for rep in 1 2 3; do
for a in "--workload big-scene" "--workload big-scene --traversal hier"; do
  python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 4 $a 2>/dev/null | line "6 waves $a"
  bash profiles/r05/with_objs.sh "$W5" python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 4 $a 2>/dev/null | line "5 waves $a"
done; done > gpurun_out/c48_waves56.txt 2>&1
a="--workload big-scene --width 3840 --height 2160 --samples 256"
python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 $a 2>/dev/null | line "6 waves $a" >> gpurun_out/c48_waves56.txt
bash profiles/r05/with_objs.sh "$W5" python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 $a 2>/dev/null | line "5 waves $a" >> gpurun_out/c48_waves56.txt
for cfg in "PORTRAYER_WAVES=4" "X=0"; do for a in "--workload big-soup --samples 64 --traversal hier" "--workload big-mesh --samples 64 --traversal hier" "--workload big-soup --samples 16"; do
  env $cfg python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "$cfg $a"
done; done >> gpurun_out/c48_waves56.txt 2>&1
cat gpurun_out/c48_waves56.txt
