# round 4, call 2: first run of the wave-uniform k-d walk (build/variants/kdw): every test with a k-d render, then a k-d-only fuzz, then the three k-d workloads
cp build/variants/kdw/libportrayer_hip.so portrayer_amd/libportrayer_hip.so
timeout 900 python -m pytest tests -m gpu -q -x -k "kd or parallel_to_an_axis or example_matches or random_scene or extreme or chain or textured or stack or golden" > gpurun_out/c02_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c02_pytest.log
FUZZ_MODES=kd timeout 600 python3 tests/fuzz_gpu_parity.py 70000 40 > gpurun_out/c02_fuzz.log 2>&1
for wl in big-scene mirror cows; do timeout 300 python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --workload $wl --traversal kd > gpurun_out/c02_kd_$wl.json 2> gpurun_out/c02_kd_$wl.err; done
