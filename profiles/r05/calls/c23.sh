# round 5, call 23: ranks sharing one GPU get a share of the resident blocks each (8 persistent grids of full size ran one after the other, each with its own tail)
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
c=d['config'].get('collective') or {}
print('%-40s %9.1f Mray/s %8.3f ms/frame  per-rank kernel ms (one frame at a time): %s  assembled==single: %s' % ('$1', d['value'], d['ms_per_step'], [round(x,3) for x in (c.get('per_rank_kernel_ms') or {}).get('one_frame_at_a_time', [])], d['config'].get('assembled_image_equals_single_gpu_render')))"; }
for n in 8 2; do
python3 bench.py --gpus $n --same-device --steps 10 --warmup 2 --no-cpu-baseline --check 2>/dev/null | line "--gpus $n --same-device"
done > gpurun_out/c23_same_device.txt 2>&1
cat gpurun_out/c23_same_device.txt
timeout 900 python3 -m pytest tests/test_gpu_multirank.py -q -m gpu 2>&1 | tail -2
