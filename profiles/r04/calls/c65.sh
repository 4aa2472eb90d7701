# round 4, call 65: pt_node on the committed tree - 8 and 2 ranks sharing the one GPU, pipelined and one frame at a time (what the host adds to a frame)
for n in 8 2; do for p in "" "--no-pipeline"; do
timeout 600 python3 bench.py --gpus $n --same-device --workload big-scene --no-cpu-baseline --no-extras --steps 10 --warmup 3 $p 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']['collective']
print('%d ranks %-14s %9.1f Mray/s %7.3f ms/frame  via %s  one frame at a time: %s' % ($n, '$p' or 'pipelined', d['value'], d['ms_per_step'], c.get('via'), json.dumps(c.get('one_frame_at_a_time'))))"
done; done > gpurun_out/c65_node.txt 2>&1
cat gpurun_out/c65_node.txt
