# round 3, call 3: the packed slab test of the wave-uniform walks (pt_slab_pk2: node planes [axis][child], v_pk_fma_f32 with scalar plane pairs)
python -m pytest tests -m gpu -q -x > gpurun_out/c03_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c03_pytest.log
bash profiles/workloads.sh --no-extras > gpurun_out/c03_workloads.log 2>&1
for wl in "big-soup --samples 64" "big-mesh --samples 64" "big-scene --share 8" "cows --traversal hier" "big-scene --width 3840 --height 2160 --samples 256"; do
python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload $wl 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-60s %9.1f Mray/s %9.2f ms/frame' % ('$wl', d['value'], d['ms_per_step']))"
done >> gpurun_out/c03_workloads.log 2>&1
python3 tests/fuzz_gpu_parity.py 3000 300 > gpurun_out/c03_fuzz.log 2>&1
