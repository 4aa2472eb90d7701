# round 4, call 6: the wave-uniform k-d walk as the in-tree build: whole suite, k-d fuzz, the three k-d workloads
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=300 > gpurun_out/c06_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c06_pytest.log
FUZZ_MODES=kd timeout 900 python3 tests/fuzz_gpu_parity.py 71000 60 > gpurun_out/c06_fuzz.log 2>&1
FUZZ_MODES=kd timeout 600 python3 tests/fuzz_gpu_parity.py 72000 20 96 64 64 > gpurun_out/c06_fuzz64.log 2>&1
for wl in big-scene mirror cows; do timeout 300 python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload $wl --traversal kd > gpurun_out/c06_kd_$wl.json 2> gpurun_out/c06_kd_$wl.err; done
