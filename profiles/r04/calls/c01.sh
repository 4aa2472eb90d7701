# round 4, call 1: the ADVICE r03 (high) fix - rays parallel to an axis on the plain per-octant kernels - new test + the -0 test; then the whole suite; KD baseline of the round
timeout 600 python -m pytest tests/test_gpu_render_parity.py -m gpu -q -x -k "parallel_to_an_axis or negative_zero" > gpurun_out/c01_pytest_new.log 2>&1; echo "pytest rc $?" >> gpurun_out/c01_pytest_new.log
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/c01_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c01_pytest.log
for wl in big-scene mirror cows; do timeout 300 python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload $wl --traversal kd > gpurun_out/c01_kd_$wl.json 2> gpurun_out/c01_kd_$wl.err; done
timeout 300 python3 bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 > gpurun_out/c01_flat.json 2> gpurun_out/c01_flat.err
