# round 4, call 24: (a) the dielectric scene with its pixels' chunks spread over several wavefronts (PORTRAYER_LANE_CHUNKS), (b) its work-item timeline,
# (c) where a cold first frame's preparation time goes (PORTRAYER_VERBOSE laps of pt_scene_upload), (d) all workloads of the round's table
run() { name=$1; shift
  env "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-54s %9.1f Mray/s %8.3f ms  %s' % ('$name', d['value'], d['ms_per_step'], d['roofline']['kernel'][5:70]))"
}
B="python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1"
run "aquarium x16 default" X=1 $B --workload aquarium >> gpurun_out/c24_aq.txt
for c in 1 2 4; do run "aquarium x16 LANE_CHUNKS=$c" PORTRAYER_LANE_CHUNKS=$c $B --workload aquarium >> gpurun_out/c24_aq.txt; done
run "aquarium x64 default" X=1 $B --workload aquarium --samples 64 >> gpurun_out/c24_aq.txt
for c in 1 2 4; do run "aquarium x64 LANE_CHUNKS=$c" PORTRAYER_LANE_CHUNKS=$c $B --workload aquarium --samples 64 >> gpurun_out/c24_aq.txt; done
run "water-glass x16 default" X=1 $B --workload water-glass >> gpurun_out/c24_aq.txt
run "water-glass x16 LANE_CHUNKS=1" PORTRAYER_LANE_CHUNKS=1 $B --workload water-glass >> gpurun_out/c24_aq.txt
bash profiles/timeline.sh "--workload aquarium" "--workload aquarium --samples 64" "--workload big-scene" > gpurun_out/c24_timeline.txt 2>&1
PORTRAYER_LANE_CHUNKS=1 bash profiles/timeline.sh "--workload aquarium" "--workload aquarium --samples 64" >> gpurun_out/c24_timeline.txt 2>&1
PORTRAYER_VERBOSE=1 python3 bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 0 --workload big-scene > gpurun_out/c24_cold.json 2> gpurun_out/c24_cold.err
bash profiles/workloads.sh > gpurun_out/c24_workloads.txt 2>&1
