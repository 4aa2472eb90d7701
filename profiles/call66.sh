python -m pytest tests -m gpu -x -q > gpurun_out/c66_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c66_pytest.log
( for w in 3 4; do echo "== PORTRAYER_WAVES=$w"; export PORTRAYER_WAVES=$w
for wl in "big-scene" "big-scene --traversal hier" "big-scene --width 3840 --height 2160 --samples 256 --steps 2" "big-scene --share 8" "big-soup --samples 64" "big-soup" "big-soup --samples 64 --traversal hier" "big-mesh" "big-mesh --samples 64" "cows" "cows --traversal hier"; do
python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload $wl 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-70s %9.1f Mray/s %9.2f ms/frame' % ('$wl', d['value'], d['ms_per_step']))"
done; done ) > gpurun_out/c66_waves.log 2>&1
unset PORTRAYER_WAVES
bash profiles/workloads.sh --no-extras > gpurun_out/c66_workloads.log 2>&1
timeout 900 python tests/fuzz_gpu_parity.py 120000 150 > gpurun_out/c66_fuzz.log 2>&1
PORTRAYER_WAVES=4 timeout 900 python tests/fuzz_gpu_parity.py 121000 150 >> gpurun_out/c66_fuzz.log 2>&1
