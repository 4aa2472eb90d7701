# round 3, call 35: the chain kernel at 4 waves per SIMD by default in the flat_scene semantics: suite, fuzz, the mirror scene's profile again
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/c35_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c35_pytest.log
bash profiles/run_profile.sh r03_mirror --workload mirror > gpurun_out/c35_prof5.log 2>&1
timeout 600 python3 tests/fuzz_gpu_parity.py 23000 80 > gpurun_out/c35_fuzz.log 2>&1
bash profiles/workloads.sh --no-extras > gpurun_out/c35_workloads.log 2>&1
