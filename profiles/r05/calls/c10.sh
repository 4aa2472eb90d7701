# round 5, call 10: one rank's 1/8 share of the headline frame, EVERY rank (only rank 0 had ever been timed), kernel time from HIP events; then the timeline of a share
for k in 0 1 2 3 4 5 6 7; do
  python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 --workload big-scene --share 8 --share-rank $k 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('share 8 rank $k: %9.1f Mray/s %8.3f ms/frame kernel %8.3f ms  rays %d' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config'].get('rays_per_frame', 0)))"
done > gpurun_out/c10_shares.txt 2>&1
python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 --workload big-scene 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('full frame    : %9.1f Mray/s %8.3f ms/frame kernel %8.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" >> gpurun_out/c10_shares.txt
cat gpurun_out/c10_shares.txt
bash profiles/timeline.sh "--workload big-scene" "--workload big-scene --share 8" "--workload big-scene --share 8 --share-rank 3" > gpurun_out/c10_timeline.txt 2>&1; cat gpurun_out/c10_timeline.txt
