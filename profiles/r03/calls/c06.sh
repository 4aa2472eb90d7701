# round 3, call 6: tree step specialised by the wavefront's direction octant (no min/max to tell entering from leaving planes)
python -m pytest tests -m gpu -q -x > gpurun_out/c06_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c06_pytest.log
bash profiles/workloads.sh --no-extras > gpurun_out/c06_workloads.log 2>&1
for wl in "big-soup --samples 64" "big-mesh --samples 64" "big-scene --share 8" "cows --traversal hier" "mirror --traversal kd"; do
python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload $wl 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-60s %9.1f Mray/s %9.2f ms/frame' % ('$wl', d['value'], d['ms_per_step']))"
done >> gpurun_out/c06_workloads.log 2>&1
bash profiles/pmc_quick.sh "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_LDS" --no-extras --workload big-scene > gpurun_out/c06_pmc.log 2>&1
python3 tests/fuzz_gpu_parity.py 7000 250 > gpurun_out/c06_fuzz.log 2>&1
python3 tests/fuzz_gpu_parity.py 8000 40 64 48 32 > gpurun_out/c06_fuzz32.log 2>&1
