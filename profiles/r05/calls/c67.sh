# round 5, call 67: a longer fuzz on the final tree (new seeds; default switches, PARK=0, FORK=1, device-built trees for every mesh), the config-size tests
timeout 1500 python3 tests/fuzz_gpu_parity.py 270000 200 > gpurun_out/c67_fuzz_a.log 2>&1; tail -1 gpurun_out/c67_fuzz_a.log
PORTRAYER_PARK=0 timeout 900 python3 tests/fuzz_gpu_parity.py 271000 60 > gpurun_out/c67_fuzz_b.log 2>&1; tail -1 gpurun_out/c67_fuzz_b.log
PORTRAYER_BUILD=device PORTRAYER_BUILD_MIN=16 timeout 900 python3 tests/fuzz_gpu_parity.py 272000 60 > gpurun_out/c67_fuzz_c.log 2>&1; tail -1 gpurun_out/c67_fuzz_c.log
PORTRAYER_FORK=1 timeout 900 python3 tests/fuzz_gpu_parity.py 273000 40 > gpurun_out/c67_fuzz_d.log 2>&1; tail -1 gpurun_out/c67_fuzz_d.log
PORTRAYER_MESH_OCT=0 timeout 900 python3 tests/fuzz_gpu_parity.py 274000 40 > gpurun_out/c67_fuzz_e.log 2>&1; tail -1 gpurun_out/c67_fuzz_e.log
