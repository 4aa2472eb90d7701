for wl in triangle primitives cows; do
python3 bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 20 --workload $wl 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-12s %9.1f Mray/s  %8.4f ms/frame wall, kernel %8.4f ms, host-buffer path %s' % ('$wl', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config'].get('host_buffer_path')))"
done > gpurun_out/c85.log 2>&1
