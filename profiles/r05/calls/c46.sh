# round 5, call 46: instance walks as a call specialised by octant (no markers, no re-dispatch per descent): parity of everything with meshes, then A/B against the marker walk (mode 1)
timeout 1800 python -m pytest tests/test_gpu_render_parity.py tests/test_gpu_switch_matrix.py tests/test_gpu_textures.py -m gpu -q -x --timeout=900 > gpurun_out/c46_pytest.log 2>&1; tail -1 gpurun_out/c46_pytest.log
line() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-72s %9.1f Mray/s %8.3f ms/frame' % ('$1', d['value'], d['ms_per_step']))"; }
for rep in 1 2; do
for a in "--workload big-soup --samples 64" "--workload big-mesh --samples 64" "--workload cows" "--workload mirror"; do
  bash profiles/r05/with_objs.sh "1=build/diag/m1_marker_walk.o" python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "marker walk $a"
  python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $a 2>/dev/null | line "instance call $a"
done; done > gpurun_out/c46_instance_call.txt 2>&1
cat gpurun_out/c46_instance_call.txt
